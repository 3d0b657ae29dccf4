// pixel.hip -- the image half of the transform-robustness path, hand-written for gfx950.
//
//   k_composite   add_noise (transforms.py:143-159) + GaussianBlur/where/clip (transforms.py:55-62)
//                 fused per 32x32 tile; one RGB f32 composite (12-byte texels) per (candidate, blur kernel).
//   k_warp_net    warpPerspective (perspective_transform.py:47) + gamma LUT (transforms.py:69-75)
//                 + resize to net x net + clamp + normalise (transforms.py:18-29), fused, and
//                 evaluated ON DEMAND: the resize reads a 2x2 neighbourhood of the warp per output
//                 pixel, so only <= 4*net*net of the (up to 275x275) warped pixels are computed.
//   k_resize, k_add_noise, k_warp_full: the same operators as stand-alone calls.
//
// Arithmetic is float32 / float64 in the exact operation order of oracle/graphite_oracle.c; this file
// is compiled with -ffp-contract=off so that no mul+add pair is fused behind its back.
// Everything here is bandwidth / gather work: no MFMA, coalesced 16-byte accesses, LDS for the
// blur stencil and the LUT.
#include "common.h"

namespace {

struct Tap { int i0, i1; float w0, w1; };

// cv::resize INTER_LINEAR tap.  scale = 1.0/((double)dsize/ssize) (host).
__device__ __forceinline__ Tap tap_h(int d, int ssize, double scale) {
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
    Tap t;
    t.i0 = s;
    t.i1 = s + 1 < ssize ? s + 1 : ssize - 1;
    t.w0 = 1.f - f;
    t.w1 = f;
    return t;
}
__device__ __forceinline__ int clipi(int x, int a, int b) { return x >= a ? (x < b ? x : b - 1) : a; }
__device__ __forceinline__ Tap tap_v(int d, int ssize, double scale) {
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    Tap t;
    t.i0 = clipi(s, 0, ssize);
    t.i1 = clipi(s + 1, 0, ssize);
    t.w0 = 1.f - f;
    t.w1 = f;
    return t;
}
__device__ __forceinline__ float bilerp(const float* __restrict__ p, int sw, const Tap& tx, const Tap& ty) {
    float h0 = p[ty.i0 * sw + tx.i0] * tx.w0 + p[ty.i0 * sw + tx.i1] * tx.w1;
    float h1 = p[ty.i1 * sw + tx.i0] * tx.w0 + p[ty.i1 * sw + tx.i1] * tx.w1;
    return h0 * ty.w0 + h1 * ty.w1;
}
__device__ __forceinline__ float clamp01(float v) { return v < 0.f ? 0.f : (v > 1.f ? 1.f : v); }
__device__ __forceinline__ int refl101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) { if (i < 0) i = -i; else i = 2 * n - 2 - i; }
    return i;
}

// ---------------------------------------------------------------------------------------------
// stand-alone resize (+ optional clamp / -0.5)
// ---------------------------------------------------------------------------------------------
__global__ void k_resize(const float* __restrict__ src, int C, int sh, int sw, float* __restrict__ dst, int dh, int dw,
                         double sx, double sy, int clamp, int normalize) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    int total = C * dh * dw;
    if (idx >= total) return;
    int c = idx / (dh * dw), r = idx % (dh * dw), y = r / dw, x = r % dw;
    const float* p = src + (size_t)c * sh * sw;
    float v;
    if (sh == dh && sw == dw) v = p[y * sw + x];
    else if (sh == 2 * dh && sw == 2 * dw) {      // cv::resize's INTER_AREA fast path for an exact 2x2 reduction
        const float* q = p + (2 * y) * sw + 2 * x;
        v = (((0.f + q[0]) + q[1]) + q[sw]) + q[sw + 1];
        v = v * 0.25f;
    }
    else v = bilerp(p, sw, tap_h(x, sw, sx), tap_v(y, sh, sy));
    if (clamp) v = clamp01(v);
    if (normalize) v = v - 0.5f;
    dst[idx] = v;
}

// ---------------------------------------------------------------------------------------------
// stand-alone add_noise
// ---------------------------------------------------------------------------------------------
__global__ void k_add_noise(const float* __restrict__ img, int H, int W, const float* __restrict__ mask,
                            const float* __restrict__ theta, int nh, int nw, double sx, double sy, float lbd, int clip,
                            float* comb, float* pert, float* mask_lg) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    int total = 3 * H * W;
    if (idx >= total) return;
    int c = idx / (H * W), r = idx % (H * W), y = r / W, x = r % W;
    Tap tx = tap_h(x, nw, sx), ty = tap_v(y, nh, sy);
    bool same = (nh == H && nw == W);
    float th = same ? theta[idx] : bilerp(theta + c * nh * nw, nw, tx, ty);
    float ml = same ? mask[idx] : bilerp(mask + c * nh * nw, nw, tx, ty);
    float v = img[idx] + lbd * th;
    float m = ml > 0.5f ? 1.f : 0.f;
    if (clip) v = clamp01(v);
    if (comb) comb[idx] = v;
    if (pert) pert[idx] = m > 0.5f ? v : 0.f;
    if (mask_lg) mask_lg[idx] = m;
}

// ---------------------------------------------------------------------------------------------
// composite: per 32x32 tile (+3 halo) of one candidate, every blur variant the table uses.
// ---------------------------------------------------------------------------------------------
constexpr int CT = 32;          // tile edge
constexpr int CHALO = 3;        // max blur radius (k=7)
constexpr int CE = CT + 2 * CHALO;   // 38
constexpr int CEP = CE + 1;     // padded row

__device__ __forceinline__ float rowf(const float* s, int k) {   // s points at the centre tap
    if (k == 3) return s[0] * 0.5f + (s[-1] + s[1]) * 0.25f;
    if (k == 5) return s[0] * 0.375f + (s[-1] + s[1]) * 0.25f + (s[-2] + s[2]) * 0.0625f;
    float v = 0.03125f * s[-3];
    v += 0.109375f * s[-2];
    v += 0.21875f * s[-1];
    v += 0.28125f * s[0];
    v += 0.21875f * s[1];
    v += 0.109375f * s[2];
    v += 0.03125f * s[3];
    return v;
}
__device__ __forceinline__ float colf(const float* s, int stride, int k) {   // s points at the centre tap
    if (k == 3) return s[0] * 0.5f + (s[-stride] + s[stride]) * 0.25f;
    if (k == 5) {
        float v = 0.375f * s[0];
        v += 0.25f * (s[stride] + s[-stride]);
        v += 0.0625f * (s[2 * stride] + s[-2 * stride]);
        return v;
    }
    float v = 0.28125f * s[0];
    v += 0.21875f * (s[stride] + s[-stride]);
    v += 0.109375f * (s[2 * stride] + s[-2 * stride]);
    v += 0.03125f * (s[3 * stride] + s[-3 * stride]);
    return v;
}

template <bool FROM_PERT>
__global__ __launch_bounds__(256) void k_composite(const float* __restrict__ img, int H, int W,
                                                   const float* __restrict__ masks, const float* __restrict__ thetas,
                                                   int nh, int nw, double sx, double sy, float lbd, int nslots, int b0,
                                                   int b1, int b2, int b3, float* __restrict__ att) {
    __shared__ float s_pert[3][CE][CEP];
    __shared__ float s_tmp[3][CE][CT];
    __shared__ unsigned char s_msk[3][CE][CEP];
    const int tiles_x = (W + CT - 1) / CT;
    const int tx0 = (blockIdx.x % tiles_x) * CT, ty0 = (blockIdx.x / tiles_x) * CT;
    const int cand = blockIdx.y;
    const int tid = threadIdx.x;
    const size_t pl = (size_t)H * W;
    const float* mask = masks + (FROM_PERT ? 0 : (size_t)cand * 3 * nh * nw);
    const float* theta = thetas + (FROM_PERT ? 0 : (size_t)cand * 3 * nh * nw);

    // (a) pert and mask bits on the haloed tile; out-of-image coordinates reflect (BORDER_REFLECT_101)
    for (int i = tid; i < CE * CE; i += 256) {
        int ly = i / CE, lx = i % CE;
        int gy = refl101(ty0 + ly - CHALO, H), gx = refl101(tx0 + lx - CHALO, W);
        if (FROM_PERT) {
            for (int c = 0; c < 3; c++) {
                s_pert[c][ly][lx] = thetas[c * pl + (size_t)gy * W + gx];           // thetas == pert here
                s_msk[c][ly][lx] = masks[c * pl + (size_t)gy * W + gx] > 0.5f;      // masks == mask_lg here
            }
        } else {
            Tap tx = tap_h(gx, nw, sx), ty = tap_v(gy, nh, sy);
            for (int c = 0; c < 3; c++) {
                float th = bilerp(theta + c * nh * nw, nw, tx, ty);
                float ml = bilerp(mask + c * nh * nw, nw, tx, ty);
                float v = clamp01(img[c * pl + (size_t)gy * W + gx] + lbd * th);
                bool m = ml > 0.5f;
                s_pert[c][ly][lx] = m ? v : 0.f;
                s_msk[c][ly][lx] = m;
            }
        }
    }
    __syncthreads();

    const int blurs[4] = {b0, b1, b2, b3};
    for (int s = 0; s < nslots; s++) {
        const int k = blurs[s];
        if (k != 0) {
            for (int i = tid; i < 3 * CE * CT; i += 256) {          // row pass on all 38 rows
                int c = i / (CE * CT), r = i % (CE * CT), ly = r / CT, x = r % CT;
                s_tmp[c][ly][x] = rowf(&s_pert[c][ly][x + CHALO], k);
            }
            __syncthreads();
        }
        float* dst = att + ((size_t)cand * nslots + s) * pl * ATT_C;
        for (int j = 0; j < 4; j++) {
            int y = tid / CT + 8 * j, x = tid % CT;
            int gy = ty0 + y, gx = tx0 + x;
            if (gy < H && gx < W) {
                float o[3];
                for (int c = 0; c < 3; c++) {
                    float v = (k == 0) ? s_pert[c][y + CHALO][x + CHALO] : colf(&s_tmp[c][y + CHALO][x], CT, k);
                    float base = img[c * pl + (size_t)gy * W + gx];
                    o[c] = clamp01(s_msk[c][y + CHALO][x + CHALO] ? v : base);
                }
                *reinterpret_cast<float3*>(dst + ((size_t)gy * W + gx) * ATT_C) = make_float3(o[0], o[1], o[2]);
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// warp sampling
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int cv_round_sat(double v) {
    v = fmax(v, -2147483648.0);
    v = fmin(v, 2147483647.0);
    return (int)rint(v);
}
__device__ __forceinline__ int sat_short(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

struct WarpTaps { int o00, o01, o10, o11; float w0, w1, w2, w3; };

// cv::warpPerspective coordinate + weight computation for dst pixel (x,y)
__device__ __forceinline__ WarpTaps warp_taps(const double* Mi, int bw0, int x, int y, int H, int W) {
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
    WarpTaps t;
    t.w0 = (1.f - fy) * (1.f - fx);
    t.w1 = (1.f - fy) * fx;
    t.w2 = fy * (1.f - fx);
    t.w3 = fy * fx;
    int sx0 = clipi(sx, 0, W), sx1 = clipi(sx + 1, 0, W), sy0 = clipi(sy, 0, H), sy1 = clipi(sy + 1, 0, H);
    t.o00 = sy0 * W + sx0;
    t.o01 = sy0 * W + sx1;
    t.o10 = sy1 * W + sx0;
    t.o11 = sy1 * W + sx1;
    return t;
}

// gamma stage (transforms.py:69-75): u8 = trunc(v*255); table lookup; /255.0 in double; clip.
// lutf[u] = clamp01((float)((double)lut[u] / 255.0)) is prepared once per workgroup.
__device__ __forceinline__ float gamma_stage_f(float v, const float* lutf) {
    float s = v * 255.0f;
    return lutf[((int)s) & 255];
}

// warped + gamma'd pixel from an RGB composite (12-byte texels)
__device__ __forceinline__ void warp_gamma_rgbx(const float3* __restrict__ att, const double* Mi, int bw0, int x, int y,
                                                int H, int W, const float* lutf, float* o3) {
    WarpTaps t = warp_taps(Mi, bw0, x, y, H, W);
    float3 a = att[t.o00], b = att[t.o01], c = att[t.o10], d = att[t.o11];
    o3[0] = gamma_stage_f(a.x * t.w0 + b.x * t.w1 + c.x * t.w2 + d.x * t.w3, lutf);
    o3[1] = gamma_stage_f(a.y * t.w0 + b.y * t.w1 + c.y * t.w2 + d.y * t.w3, lutf);
    o3[2] = gamma_stage_f(a.z * t.w0 + b.z * t.w1 + c.z * t.w2 + d.z * t.w3, lutf);
}

// One workgroup per image n = (candidate, transform).
// xcd_map: workgroups are dealt round-robin to the 8 XCDs (each with its own L2); with the map on, XCD x gets the
// x-th contiguous eighth of the launch's images, i.e. (for T ~ chunk/8) the images of ONE candidate, whose composites
// then stay in that XCD's L2 instead of every L2 streaming every candidate's composites.
__device__ __forceinline__ int warp_local_index(int count, int xcd_map) {
    if (!xcd_map) return blockIdx.x;
    const int per = (count + 7) >> 3;
    return (blockIdx.x & 7) * per + (blockIdx.x >> 3);   // >= count for the padding workgroups of the rounded-up grid
}

// convert2Network's size rule for net_size != 32 (transforms.py:35-40, the ALPR branch): the SHORT side of the warped
// image becomes net_size, the long side int(round(net_size / short * long)) -- Python float arithmetic, round half even.
__host__ __device__ inline void net_size_rule(int cw, int ch, int net_size, int* nx, int* ny) {
    if (ch > cw) { *nx = net_size; *ny = (int)rint((double)net_size / (double)cw * (double)ch); }
    else { *ny = net_size; *nx = (int)rint((double)net_size / (double)ch * (double)cw); }
}

// Generic network size, CHW output: every warped pixel runs the full coordinate computation.
// var == 0: net x net output, `normalize` as given (the net_size == 32 branch of convert2Network for any net).
// var == 1: the net_size != 32 branch: per transform ny x nx by net_size_rule, clamp only (that branch never normalises);
//           image `local` is written at out + local * out_stride as [3, ny, nx].
__global__ __launch_bounds__(256) void k_warp_net_any(const float* __restrict__ att_all, int H, int W, int nslots,
                                                      const XformDev* __restrict__ xf, int T, long long n_begin, int cand0,
                                                      int normalize, int net, int var, long long out_stride, int count,
                                                      int xcd_map, float* __restrict__ out) {
    __shared__ double s_Mi[9];
    __shared__ float s_lutf[256];
    __shared__ int s_par[4];
    const int local = warp_local_index(count, xcd_map);
    if (local >= count) return;                   // uniform for the workgroup
    const long long n = n_begin + local;
    const int cand = (int)(n / T) - cand0, t = (int)(n % T);
    const XformDev* x = xf + t;
    const int tid = threadIdx.x;
    if (tid < 9) s_Mi[tid] = x->Mi[tid];
    s_lutf[tid] = clamp01((float)((double)x->lut[tid] / 255.0));
    if (tid == 128) { s_par[0] = x->crop_x; s_par[1] = x->crop_y; s_par[2] = x->slot; s_par[3] = x->bw0; }
    __syncthreads();
    const int cw = s_par[0], ch = s_par[1], bw0 = s_par[3];
    const size_t pl = (size_t)H * W;
    const float3* att = reinterpret_cast<const float3*>(att_all) + ((size_t)cand * nslots + s_par[2]) * pl;
    int nx = net, ny = net;
    if (var) { net_size_rule(cw, ch, net, &nx, &ny); normalize = 0; }
    const double scx = 1.0 / ((double)nx / (double)cw), scy = 1.0 / ((double)ny / (double)ch);
    const bool same = (cw == nx && ch == ny);
    const bool area2 = (cw == 2 * nx && ch == 2 * ny);
    float* o = out + (size_t)local * (size_t)out_stride;
    const int npx = nx * ny;
    for (int p = tid; p < npx; p += 256) {
        int oy = p / nx, ox = p % nx;
        float r[3];
        if (same) {
            warp_gamma_rgbx(att, s_Mi, bw0, ox, oy, H, W, s_lutf, r);
        } else {
            Tap tx = tap_h(ox, cw, scx), ty = tap_v(oy, ch, scy);
            float g00[3], g01[3], g10[3], g11[3];
            warp_gamma_rgbx(att, s_Mi, bw0, tx.i0, ty.i0, H, W, s_lutf, g00);
            warp_gamma_rgbx(att, s_Mi, bw0, tx.i1, ty.i0, H, W, s_lutf, g01);
            warp_gamma_rgbx(att, s_Mi, bw0, tx.i0, ty.i1, H, W, s_lutf, g10);
            warp_gamma_rgbx(att, s_Mi, bw0, tx.i1, ty.i1, H, W, s_lutf, g11);
            for (int c = 0; c < 3; c++) {
                if (area2) {                      // exact 2x2 reduction: cv::resize's INTER_AREA fast path
                    r[c] = ((((0.f + g00[c]) + g01[c]) + g10[c]) + g11[c]) * 0.25f;
                } else {
                    float h0 = g00[c] * tx.w0 + g01[c] * tx.w1;
                    float h1 = g10[c] * tx.w0 + g11[c] * tx.w1;
                    r[c] = h0 * ty.w0 + h1 * ty.w1;
                }
            }
        }
        for (int c = 0; c < 3; c++) {
            float v = clamp01(r[c]);
            o[c * npx + p] = normalize ? v - 0.5f : v;
        }
    }
}

// The 32x32 network input (every BASELINE configuration), fused with the classifiers' input stage.
// A transform of the reference is a rotation about the image's y axis (perspective_transform.py:16-29), so the inverse
// matrix has Mi[1] == Mi[7] == 0: the projective divide, the source x and its 1/32-pixel fraction depend on the
// destination COLUMN only.  The 64 warped columns an image needs (two per output column) are worked out ONCE per image
// in double precision -- the expensive part, an IEEE f64 division per warped pixel, drops from 4096 to 64 per image --
// and kept in LDS; what remains per warped pixel is the row coordinate (two f64 adds, one multiply, one rounding).
// Every intermediate is computed by the same operations in the same order as warp_taps() (adding Mi[1]*y = +-0 or
// Mi[7]*y = +-0 changes no value), so the results stay bit-identical; transforms whose matrix is not of that form
// (sep == 0) take the per-pixel path inside the same kernel.
struct WarpCol {
    double Wd;          // 32 / W of the column (0 when W == 0)
    double A, B;        // Mi[3]*xb and Mi[3]*x1: the column's share of Y0
    int sx0, sx1;       // clipped source columns
    float fx;           // fraction (X & 31) / 32
    int pad_;
};

template <int OUT>
__global__ __launch_bounds__(256) void k_warp_net32(const float* __restrict__ att_all, int H, int W, int nslots,
                                                    const XformDev* __restrict__ xf, int T, long long n_begin, int cand0,
                                                    int normalize, int count, int xcd_map, const float* __restrict__ c11,
                                                    float* __restrict__ out) {
    constexpr int net = 32;
    __shared__ double s_Mi[9];
    __shared__ float s_lutf[256];
    __shared__ WarpCol s_col[64];
    __shared__ Tap s_tx[32], s_ty[32];
    __shared__ float s_c11[12];
    const int local = warp_local_index(count, xcd_map);
    if (local >= count) return;                   // uniform for the workgroup
    const long long n = n_begin + local;
    const int cand = (int)(n / T) - cand0, t = (int)(n % T);
    const XformDev* x = xf + t;
    const int tid = threadIdx.x;
    const int cw = x->crop_x, ch = x->crop_y, bw0 = x->bw0, sep = x->sep;
    const bool same = (cw == net && ch == net);
    const bool area2 = (cw == 2 * net && ch == 2 * net);      // exact 2x2 reduction (crop 64): taps 2o, 2o+1 as bilinear gives
    const double scx = 1.0 / ((double)net / (double)cw), scy = 1.0 / ((double)net / (double)ch);
    s_lutf[tid] = clamp01((float)((double)x->lut[tid] / 255.0));
    if (tid < 9) s_Mi[tid] = x->Mi[tid];
    if (OUT == GR_OUT_NHWC4_C11 && tid >= 16 && tid < 28) s_c11[tid - 16] = c11[tid - 16];
    if (tid >= 64 && tid < 96) {
        const int o = tid - 64;
        Tap tx, ty;
        if (same) { tx.i0 = tx.i1 = ty.i0 = ty.i1 = o; tx.w0 = ty.w0 = 1.f; tx.w1 = ty.w1 = 0.f; }
        else { tx = tap_h(o, cw, scx); ty = tap_v(o, ch, scy); }
        s_tx[o] = tx; s_ty[o] = ty;
    }
    if (tid < 64 && sep) {                        // warped column (tid & 1 ? i1 : i0) of output column tid >> 1
        const int ox = tid >> 1;
        int xc;
        if (same) xc = ox;
        else { Tap tx = tap_h(ox, cw, scx); xc = (tid & 1) ? tx.i1 : tx.i0; }
        const double m0 = x->Mi[0], m2 = x->Mi[2], m3 = x->Mi[3], m6 = x->Mi[6], m8 = x->Mi[8];
        const int xb = (xc / bw0) * bw0, x1 = xc - xb;
        const double X0 = m0 * xb + m2;           // (Mi[0]*xb + Mi[1]*y) + Mi[2] with Mi[1]*y == +-0
        const double W0 = m6 * xb + m8;           // (Mi[6]*xb + Mi[7]*y) + Mi[8] with Mi[7]*y == +-0
        double Wd = W0 + m6 * x1;
        Wd = 32.0 * (Wd != 0.0 ? 1.0 / Wd : 0.0);
        const int X = cv_round_sat((X0 + m0 * x1) * Wd);
        const int sx = sat_short(X >> 5);
        WarpCol c;
        c.Wd = Wd; c.A = m3 * xb; c.B = m3 * x1;
        c.sx0 = clipi(sx, 0, W); c.sx1 = clipi(sx + 1, 0, W);
        c.fx = (float)(X & 31) * (1.f / 32.f);
        c.pad_ = 0;
        s_col[tid] = c;
    }
    __syncthreads();
    const size_t pl = (size_t)H * W;
    const float3* att = reinterpret_cast<const float3*>(att_all) + ((size_t)cand * nslots + x->slot) * pl;
    const double m4 = s_Mi[4], m5 = s_Mi[5];
#pragma unroll 1
    for (int p = tid; p < net * net; p += 256) {
        const int oy = p >> 5, ox = p & 31;
        const Tap tx = s_tx[ox], ty = s_ty[oy];
        float g[2][2][3];
        if (sep) {
            // straight-line 2 x 2 (a `same` image has coinciding taps with weights 1 and 0: the single sample, exactly), so
            // that the coordinate chains and the 16 gathers of an output pixel are independent and overlap
            const WarpCol cA = s_col[ox * 2], cB = s_col[ox * 2 + 1];
            const double my[2] = {m4 * ty.i0, m4 * ty.i1};
#pragma unroll
            for (int jy = 0; jy < 2; jy++) {
                float3 tp[2][4];
                float w[2][4];
#pragma unroll
                for (int jx = 0; jx < 2; jx++) {
                    const WarpCol& c = jx ? cB : cA;
                    const double Y0 = (c.A + my[jy]) + m5;
                    const int Y = cv_round_sat((Y0 + c.B) * c.Wd);
                    const int sy = sat_short(Y >> 5);
                    const float fy = (float)(Y & 31) * (1.f / 32.f), fx = c.fx;
                    w[jx][0] = (1.f - fy) * (1.f - fx); w[jx][1] = (1.f - fy) * fx; w[jx][2] = fy * (1.f - fx); w[jx][3] = fy * fx;
                    const int r0 = clipi(sy, 0, H) * W, r1 = clipi(sy + 1, 0, H) * W;
                    tp[jx][0] = att[r0 + c.sx0]; tp[jx][1] = att[r0 + c.sx1]; tp[jx][2] = att[r1 + c.sx0]; tp[jx][3] = att[r1 + c.sx1];
                }
#pragma unroll
                for (int jx = 0; jx < 2; jx++) {
                    g[jy][jx][0] = gamma_stage_f(tp[jx][0].x * w[jx][0] + tp[jx][1].x * w[jx][1] + tp[jx][2].x * w[jx][2] + tp[jx][3].x * w[jx][3], s_lutf);
                    g[jy][jx][1] = gamma_stage_f(tp[jx][0].y * w[jx][0] + tp[jx][1].y * w[jx][1] + tp[jx][2].y * w[jx][2] + tp[jx][3].y * w[jx][3], s_lutf);
                    g[jy][jx][2] = gamma_stage_f(tp[jx][0].z * w[jx][0] + tp[jx][1].z * w[jx][1] + tp[jx][2].z * w[jx][2] + tp[jx][3].z * w[jx][3], s_lutf);
                }
            }
        } else {
#pragma unroll
            for (int jy = 0; jy < 2; jy++)
#pragma unroll
                for (int jx = 0; jx < 2; jx++)
                    warp_gamma_rgbx(att, s_Mi, bw0, jx ? tx.i1 : tx.i0, jy ? ty.i1 : ty.i0, H, W, s_lutf, g[jy][jx]);
        }
        float v[3];
        for (int c = 0; c < 3; c++) {
            float r;
            if (same) r = g[0][0][c];
            else if (area2) r = ((((0.f + g[0][0][c]) + g[0][1][c]) + g[1][0][c]) + g[1][1][c]) * 0.25f;   // INTER_AREA fast path
            else {
                const float h0 = g[0][0][c] * tx.w0 + g[0][1][c] * tx.w1;
                const float h1 = g[1][0][c] * tx.w0 + g[1][1][c] * tx.w1;
                r = h0 * ty.w0 + h1 * ty.w1;
            }
            r = clamp01(r);
            v[c] = normalize ? r - 0.5f : r;
        }
        if (OUT == GR_OUT_CHW) {
            float* o = out + (size_t)local * 3 * net * net;
            for (int c = 0; c < 3; c++) o[c * net * net + p] = v[c];
        } else {
            float4 o4;
            if (OUT == GR_OUT_NHWC4_C11) {        // GTSRBNet conv1_1 (1x1) + ReLU, the arithmetic of k_input_nhwc4<true>
                float r[3];
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    float a = 0.f;
                    a = fmaf(v[0], s_c11[k * 3 + 0], a);
                    a = fmaf(v[1], s_c11[k * 3 + 1], a);
                    a = fmaf(v[2], s_c11[k * 3 + 2], a);
                    a = a + s_c11[9 + k];
                    r[k] = a > 0.f ? a : 0.f;
                }
                o4 = make_float4(r[0], r[1], r[2], 0.f);
            } else {
                o4 = make_float4(v[0], v[1], v[2], 0.f);
            }
            reinterpret_cast<float4*>(out)[(size_t)local * (net * net) + p] = o4;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Quad-gather warp (the default for launches whose transforms are all separable).
// k_warp_net32 gives a lane one OUTPUT pixel and lets it fetch its 16 texels in 16 instructions: in each of them the 64
// lanes read 64 different 128-byte lines (the source columns of neighbouring output pixels are ~122 bytes apart), and the
// texture path looks lines up about one per clock -- 64 clocks per instruction, 16.4 K clocks per image, TA busy 90 %
// (profiles/r02_final_pmc_ta_trhits.csv).  Here a lane is one TEXEL COLUMN of one warped pixel: the four lanes of a quad
// hold (warped column 2*ox, source column sx0 / sx1) and (2*ox+1, sx0 / sx1) -- 80-odd contiguous bytes of one source
// row -- and a wavefront covers 16 neighbouring output pixels, so that one load instruction reads one contiguous ~2 KB run
// of a source row (16-17 lines instead of 64).  Each lane fetches the top and the bottom texel of its column for both
// warped rows (4 loads), multiplies by its own weights and the lane pair adds the products up over DPP in the order of
// warp_gamma_rgbx -- ((a*w0 + b*w1) + c*w2) + d*w3, float addition being commutative --, the even lane applies the gamma
// table, quad lane 0 fetches the neighbouring warped column's value over DPP and finishes the pixel.
// The row coordinates (the f64 part: Y of each of the 64 x 64 warped pixels) are worked out once per image into LDS.
// Arithmetic: k_warp_net32's separable path, value for value.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float dpp_pair_swap(float v) {          // quad_perm [1,0,3,2]
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_quad_lane2(float v) {         // quad_perm [2,2,2,2]
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xAA, 0xf, 0xf, true));
}

template <int OUT>
__global__ __launch_bounds__(256) void k_warp_quad(const float* __restrict__ att_all, int H, int W, int nslots,
                                                   const XformDev* __restrict__ xf, int T, long long n_begin, int cand0,
                                                   int normalize, int count, int xcd_map, const float* __restrict__ c11,
                                                   float* __restrict__ out) {
    constexpr int net = 32;
    __shared__ float s_lutf[256];
    __shared__ WarpCol s_col[64];
    __shared__ Tap s_tx[32], s_ty[32];
    __shared__ float s_c11[12];
    __shared__ int s_row[64 * 64];                // [warped row 2*oy + jy][warped column 2*ox + jx]: r0 | fy5 << 16 | (r1 - r0) << 21
    const int local = warp_local_index(count, xcd_map);
    if (local >= count) return;                   // uniform for the workgroup
    const long long n = n_begin + local;
    const int cand = (int)(n / T) - cand0, t = (int)(n % T);
    const XformDev* x = xf + t;
    const int tid = threadIdx.x;
    const int cw = x->crop_x, ch = x->crop_y, bw0 = x->bw0;
    const bool same = (cw == net && ch == net);
    const bool area2 = (cw == 2 * net && ch == 2 * net);
    const double scx = 1.0 / ((double)net / (double)cw), scy = 1.0 / ((double)net / (double)ch);
    s_lutf[tid] = clamp01((float)((double)x->lut[tid] / 255.0));
    if (OUT == GR_OUT_NHWC4_C11 && tid >= 96 && tid < 108) s_c11[tid - 96] = c11[tid - 96];
    if (tid >= 64 && tid < 96) {
        const int o = tid - 64;
        Tap tx, ty;
        if (same) { tx.i0 = tx.i1 = ty.i0 = ty.i1 = o; tx.w0 = ty.w0 = 1.f; tx.w1 = ty.w1 = 0.f; }
        else { tx = tap_h(o, cw, scx); ty = tap_v(o, ch, scy); }
        s_tx[o] = tx; s_ty[o] = ty;
    }
    if (tid < 64) {                               // warped column (tid & 1 ? i1 : i0) of output column tid >> 1, as k_warp_net32
        const int ox = tid >> 1;
        int xc;
        if (same) xc = ox;
        else { Tap tx = tap_h(ox, cw, scx); xc = (tid & 1) ? tx.i1 : tx.i0; }
        const double m0 = x->Mi[0], m2 = x->Mi[2], m3 = x->Mi[3], m6 = x->Mi[6], m8 = x->Mi[8];
        const int xb = (xc / bw0) * bw0, x1 = xc - xb;
        const double X0 = m0 * xb + m2;
        const double W0 = m6 * xb + m8;
        double Wd = W0 + m6 * x1;
        Wd = 32.0 * (Wd != 0.0 ? 1.0 / Wd : 0.0);
        const int X = cv_round_sat((X0 + m0 * x1) * Wd);
        const int sx = sat_short(X >> 5);
        WarpCol c;
        c.Wd = Wd; c.A = m3 * xb; c.B = m3 * x1;
        c.sx0 = clipi(sx, 0, W); c.sx1 = clipi(sx + 1, 0, W);
        c.fx = (float)(X & 31) * (1.f / 32.f);
        c.pad_ = 0;
        s_col[tid] = c;
    }
    __syncthreads();
    {                                             // row table: thread = warped column tid & 63, 16 of the 64 warped rows
        const int k = tid & 63;
        const double cA = s_col[k].A, cB = s_col[k].B, cWd = s_col[k].Wd;
        const double m4 = x->Mi[4], m5 = x->Mi[5];
#pragma unroll 4
        for (int i = 0; i < 16; i++) {
            const int m = (tid >> 6) + 4 * i;
            const Tap ty = s_ty[m >> 1];
            const int yr = (m & 1) ? ty.i1 : ty.i0;
            const double my = m4 * yr;
            const double Y0 = (cA + my) + m5;
            const int Y = cv_round_sat((Y0 + cB) * cWd);
            const int sy = sat_short(Y >> 5);
            const int r0 = clipi(sy, 0, H), r1 = clipi(sy + 1, 0, H);
            s_row[m * 64 + k] = r0 | ((Y & 31) << 16) | ((r1 - r0) << 21);
        }
    }
    __syncthreads();
    const size_t pl = (size_t)H * W;
    const char* att = reinterpret_cast<const char*>(att_all) + ((size_t)cand * nslots + x->slot) * pl * ATT_B;
    const unsigned Wb = (unsigned)W * (unsigned)ATT_B;   // 32-bit byte offsets from a uniform base (the launcher checks the frame size)
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = lane & 1, jx = (lane >> 1) & 1;
    const int ox = (wv & 1) * 16 + (lane >> 2);
    const int k = 2 * ox + jx;
    const unsigned sxb = (unsigned)ATT_B * (unsigned)(hh ? s_col[k].sx1 : s_col[k].sx0);
    const float fx = s_col[k].fx;
    const float wx = hh ? fx : 1.f - fx;
    const Tap tx = s_tx[ox];
#pragma unroll 2
    for (int i = 0; i < 16; i++) {
        const int oy = (wv >> 1) + 2 * i;
        const Tap ty = s_ty[oy];
        float g[2][3];
        float3 tt[2], tb[2];
        float wt[2], wb[2];
#pragma unroll
        for (int jy = 0; jy < 2; jy++) {
            const int e = s_row[(2 * oy + jy) * 64 + k];
            const unsigned ot = __umul24((unsigned)e & 0xffffu, Wb) + sxb;
            const unsigned ob = ot + ((e >> 21) ? Wb : 0u);
            const float fy = (float)((e >> 16) & 31) * (1.f / 32.f);
            wt[jy] = (1.f - fy) * wx;             // w0 (even lane) / w1 (odd lane)
            wb[jy] = fy * wx;                     // w2 / w3
            tt[jy] = *reinterpret_cast<const float3*>(att + ot);
            tb[jy] = *reinterpret_cast<const float3*>(att + ob);
        }
#pragma unroll
        for (int jy = 0; jy < 2; jy++) {
            const float pt[3] = {tt[jy].x * wt[jy], tt[jy].y * wt[jy], tt[jy].z * wt[jy]};
            const float pb[3] = {tb[jy].x * wb[jy], tb[jy].y * wb[jy], tb[jy].z * wb[jy]};
#pragma unroll
            for (int c = 0; c < 3; c++) {
                float sacc = pt[c] + dpp_pair_swap(pt[c]);
                sacc = sacc + pb[c];
                sacc = sacc + dpp_pair_swap(pb[c]);
                g[jy][c] = gamma_stage_f(sacc, s_lutf);   // the even lane's value is the warped pixel; odd lanes are ignored
            }
        }
        float v[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float g00 = g[0][c], g10 = g[1][c];
            const float g01 = dpp_quad_lane2(g00), g11 = dpp_quad_lane2(g10);
            float r;
            if (same) r = g00;
            else if (area2) r = ((((0.f + g00) + g01) + g10) + g11) * 0.25f;
            else {
                const float h0 = g00 * tx.w0 + g01 * tx.w1;
                const float h1 = g10 * tx.w0 + g11 * tx.w1;
                r = h0 * ty.w0 + h1 * ty.w1;
            }
            r = clamp01(r);
            v[c] = normalize ? r - 0.5f : r;
        }
        if ((lane & 3) == 0) {
            const int p = oy * net + ox;
            if (OUT == GR_OUT_CHW) {
                float* o = out + (size_t)local * 3 * net * net;
                for (int c = 0; c < 3; c++) o[c * net * net + p] = v[c];
            } else {
                float4 o4;
                if (OUT == GR_OUT_NHWC4_C11) {
                    float r[3];
#pragma unroll
                    for (int kk = 0; kk < 3; kk++) {
                        float a = 0.f;
                        a = fmaf(v[0], s_c11[kk * 3 + 0], a);
                        a = fmaf(v[1], s_c11[kk * 3 + 1], a);
                        a = fmaf(v[2], s_c11[kk * 3 + 2], a);
                        a = a + s_c11[9 + kk];
                        r[kk] = a > 0.f ? a : 0.f;
                    }
                    o4 = make_float4(r[0], r[1], r[2], 0.f);
                } else {
                    o4 = make_float4(v[0], v[1], v[2], 0.f);
                }
                reinterpret_cast<float4*>(out)[(size_t)local * (net * net) + p] = o4;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// LDS-band warp: the north star's "LDS-staged source tiles".  OPT-IN (GRAPHITE_WARP_BAND=1): correct, bit-identical to
// k_warp_net32, and slower than it on MI355X -- kept as the measured alternative (profiles/r02_warp_kernels.md).
//
// k_warp_net32 gathers its 16 taps per output pixel through L1: one image touches about half of the 128-byte lines of
// its 952 KB composite and no line is reused, so a launch moves ~500 KB per image from L2 to L1 (profiles/README.md:
// L2 -> L1 traffic is what bounds that kernel, not HBM and not the ALUs).  The source x of a destination column does
// not depend on the row (WarpCol), so the taps of an output COLUMN live in a few adjacent source columns.  This
// kernel turns the loop around: a workgroup takes WARP_BAND_K transforms that sample the SAME composite (candidate,
// blur slot), walks the composite in vertical bands of BW source columns staged in LDS with coalesced 16-byte loads
// (the next band is prefetched into registers while the current one is sampled), and in each band computes, for every one
// of its images, the output columns whose taps fall inside the band.  A band is read once per K images: ~110 KB per
// image instead of ~500 KB, and the gathers hit LDS.  Output columns whose footprint is wider than the band overlap
// (HALO = BW - S columns; none in the reference's parameter ranges) are sampled from global memory instead.
// Arithmetic: exactly k_warp_net32's separable path, value for value.
// ---------------------------------------------------------------------------------------------
constexpr int WB_NT = 1024;
constexpr int WB_PRE = 8;                         // float4 prefetch registers per thread: H * BW <= WB_NT * WB_PRE

struct BandCol {                                  // WarpCol, packed: A = Mi[3]*xb and B = Mi[3]*x1 are recomputed
    double Wd;
    short xb, x1, sx0, sx1;
    float fx;
    int pad_;
};
struct BandImg {                                  // per image of the workgroup
    double m3, m4, m5;
    int local, flags;                             // flags: 1 = same, 2 = area2; local < 0: image not in this launch
    float txw0[32], txw1[32];                     // horizontal resize weights per output column
    Tap ty[32];                                   // vertical resize taps per output row
};
constexpr size_t WB_TAIL = sizeof(BandCol) * WARP_BAND_K * 64 + sizeof(BandImg) * WARP_BAND_K + 1024 /* f255 */ +
                           WARP_BAND_K * 256 /* lut8 */ + WARP_BAND_K * 32 /* band_of */ + 2 * WARP_BAND_K * 32 /* list */ +
                           64 /* c11, counter */;

template <int OUT>
__global__ __launch_bounds__(WB_NT) void k_warp_band(const float* __restrict__ att_all, int H, int W, int nslots,
                                                     const XformDev* __restrict__ xf, const int* __restrict__ order,
                                                     const WarpGroup* __restrict__ groups, int G, int T, long long n_begin,
                                                     int count, int cand_lo, int ncand, int cand0, int normalize, int BW, int S,
                                                     int nb, const float* __restrict__ c11, float* __restrict__ out) {
    constexpr int K = WARP_BAND_K, net = 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_wb[];
    float4* band = reinterpret_cast<float4*>(smem_wb);                         // [H][BW], BW odd: rows land on distinct banks
    unsigned char* tail = smem_wb + (size_t)H * BW * 16;
    BandCol* cols = reinterpret_cast<BandCol*>(tail);                          // [K][64]
    BandImg* imgs = reinterpret_cast<BandImg*>(tail + sizeof(BandCol) * K * 64);   // [K]
    float* f255 = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(imgs) + sizeof(BandImg) * K);   // [256]
    unsigned char* lut8 = reinterpret_cast<unsigned char*>(f255 + 256);        // [K][256]
    unsigned char* band_of = lut8 + K * 256;                                   // [K][32]
    unsigned short* list = reinterpret_cast<unsigned short*>(band_of + K * 32);   // [K*32] (image, column) pairs of a band
    float* s_c11 = reinterpret_cast<float*>(list + K * 32);                    // [12]
    int* s_cnt = reinterpret_cast<int*>(s_c11 + 12);

    // workgroup -> (candidate, group); all groups of a candidate on one XCD (blockIdx % 8), whose L2 then holds its composites
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int cl = (jb / G) * 8 + xcd;
    if (cl >= ncand) return;
    const WarpGroup g = groups[jb % G];
    const int cand = cand_lo + cl;
    const int tid = threadIdx.x;
    const size_t pl = (size_t)H * W;
    const float3* att = reinterpret_cast<const float3*>(att_all) + ((size_t)(cand - cand0) * nslots + g.slot) * pl;

    // ---- per-image tables ----
    {
        const int k = tid >> 6, e = tid & 63;
        const int t = k < g.n ? order[g.first + k] : -1;
        const long long n = (long long)cand * T + (t >= 0 ? t : 0);
        const bool live = t >= 0 && n >= n_begin && n < n_begin + count;
        if (live) {
            const XformDev* x = xf + t;
            const int cw = x->crop_x, ch = x->crop_y, bw0 = x->bw0;
            const bool same = (cw == net && ch == net);
            const double scx = 1.0 / ((double)net / (double)cw), scy = 1.0 / ((double)net / (double)ch);
            const int ox = e >> 1;
            Tap tx;
            if (same) { tx.i0 = tx.i1 = ox; tx.w0 = 1.f; tx.w1 = 0.f; } else tx = tap_h(ox, cw, scx);
            const int xc = (e & 1) ? tx.i1 : tx.i0;
            const double m0 = x->Mi[0], m2 = x->Mi[2], m6 = x->Mi[6], m8 = x->Mi[8];
            const int xb = (xc / bw0) * bw0, x1 = xc - xb;
            const double X0 = m0 * xb + m2;
            const double W0 = m6 * xb + m8;
            double Wd = W0 + m6 * x1;
            Wd = 32.0 * (Wd != 0.0 ? 1.0 / Wd : 0.0);
            const int X = cv_round_sat((X0 + m0 * x1) * Wd);
            const int sx = sat_short(X >> 5);
            BandCol c;
            c.Wd = Wd; c.xb = (short)xb; c.x1 = (short)x1;
            c.sx0 = (short)clipi(sx, 0, W); c.sx1 = (short)clipi(sx + 1, 0, W);
            c.fx = (float)(X & 31) * (1.f / 32.f);
            c.pad_ = 0;
            cols[k * 64 + e] = c;
            BandImg* im = imgs + k;
            if (!(e & 1)) { im->txw0[ox] = tx.w0; im->txw1[ox] = tx.w1; }
            if (e < 32) {
                Tap ty;
                if (same) { ty.i0 = ty.i1 = e; ty.w0 = 1.f; ty.w1 = 0.f; } else ty = tap_v(e, ch, scy);
                im->ty[e] = ty;
            }
            if (e == 63) {
                im->m3 = x->Mi[3]; im->m4 = x->Mi[4]; im->m5 = x->Mi[5];
                im->local = (int)(n - n_begin);
                im->flags = (same ? 1 : 0) | ((cw == 2 * net && ch == 2 * net) ? 2 : 0);
            }
            reinterpret_cast<unsigned int*>(lut8 + k * 256)[e] = reinterpret_cast<const unsigned int*>(x->lut)[e];
        } else if (e == 63) {
            imgs[k].local = -1;
        }
        if (tid < 256) f255[tid] = clamp01((float)((double)tid / 255.0));
        if (OUT == GR_OUT_NHWC4_C11 && tid >= 256 && tid < 268) s_c11[tid - 256] = c11[tid - 256];
    }
    __syncthreads();
    if (tid < K * 32) {                           // which band samples output column ox of image k
        const int k = tid >> 5, ox = tid & 31;
        unsigned char b = 254;                    // 254: no such image
        if (imgs[k].local >= 0) {
            const BandCol c0 = cols[k * 64 + 2 * ox], c1 = cols[k * 64 + 2 * ox + 1];
            int lo = c0.sx0 < c1.sx0 ? c0.sx0 : c1.sx0, hi = c0.sx1 > c1.sx1 ? c0.sx1 : c1.sx1;
            if (c0.sx1 > hi) hi = c0.sx1;
            if (c1.sx0 < lo) lo = c1.sx0;
            int bb = lo / S;
            if (bb > nb - 1) bb = nb - 1;
            b = (lo >= bb * S && hi < bb * S + BW) ? (unsigned char)bb : 255;     // 255: footprint wider than a band
        }
        band_of[tid] = b;
    }

    // ---- bands ----
    const int npx = H * BW;                       // texels of a band (host guarantees npx <= WB_NT * WB_PRE)
    float4 pre[WB_PRE];
    auto fetch = [&](int b) {
        const int c0 = b * S;
#pragma unroll
        for (int u = 0; u < WB_PRE; u++) {
            const int i = tid + u * WB_NT;
            const int row = i / BW, col = i - row * BW;
            pre[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < npx && c0 + col < W) {            // 12-byte texel in HBM, 16-byte slot in LDS
                const float3 t3 = att[(size_t)row * W + c0 + col];
                pre[u] = make_float4(t3.x, t3.y, t3.z, 0.f);
            }
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int u = 0; u < WB_PRE; u++) {
            const int i = tid + u * WB_NT;
            if (i < npx) band[i] = pre[u];
        }
    };
    fetch(0);
    stash();
    if (tid == 0) *s_cnt = 0;
    __syncthreads();
    for (int b = 0; b < nb; b++) {
        if (b + 1 < nb) fetch(b + 1);             // in flight while this band is sampled
        // the (image, column) pairs of this band, compacted so that all 16 wavefronts share them evenly
        if (tid < K * 32) {
            const int bo = band_of[tid];
            if (bo == b || (bo == 255 && b == 0)) {
                const int idx = atomicAdd(s_cnt, 1);
                list[idx] = (unsigned short)(tid | (bo == 255 ? 0x8000 : 0));
            }
        }
        __syncthreads();
        const int nitems = *s_cnt * 32;
        const int c0 = b * S;
#pragma unroll 1
        for (int item = tid; item < nitems; item += WB_NT) {
            const int ent = list[item >> 5], oy = item & 31;                  // a half-wave = the 32 rows of one column
            const bool from_global = (ent & 0x8000) != 0;
            const int k = (ent & 0x7fff) >> 5, ox = ent & 31;
            const BandImg* im = imgs + k;
            const int flags = im->flags;
            const bool same = flags & 1, area2 = flags & 2;
            const Tap ty = im->ty[oy];
            const double m3 = im->m3, m4 = im->m4, m5 = im->m5;
            const unsigned char* lut = lut8 + k * 256;
            // All four warped pixels of the output pixel are evaluated in straight-line code (for a `same` image the two
            // taps of an axis coincide and carry weights 1 and 0, which reproduces the single sample exactly): the
            // coordinate chains, the 16 band reads and the 12 table look-ups are independent, so their latencies overlap.
            const BandCol cA = cols[k * 64 + 2 * ox], cB = cols[k * 64 + 2 * ox + 1];
            const double ax[2] = {m3 * (int)cA.xb, m3 * (int)cB.xb}, bx[2] = {m3 * (int)cA.x1, m3 * (int)cB.x1};
            const double wd[2] = {cA.Wd, cB.Wd};
            const float fxs[2] = {cA.fx, cB.fx};
            const int sx0[2] = {cA.sx0, cB.sx0}, sx1[2] = {cA.sx1, cB.sx1};
            const double my[2] = {m4 * ty.i0, m4 * ty.i1};
            float gq[2][2][3];
#pragma unroll
            for (int jy = 0; jy < 2; jy++) {
                int r0[2], r1[2];
                float w[2][4];
#pragma unroll
                for (int jx = 0; jx < 2; jx++) {
                    const double Y0 = (ax[jx] + my[jy]) + m5;
                    const int Y = cv_round_sat((Y0 + bx[jx]) * wd[jx]);
                    const int sy = sat_short(Y >> 5);
                    const float fy = (float)(Y & 31) * (1.f / 32.f), fx = fxs[jx];
                    w[jx][0] = (1.f - fy) * (1.f - fx); w[jx][1] = (1.f - fy) * fx; w[jx][2] = fy * (1.f - fx); w[jx][3] = fy * fx;
                    r0[jx] = clipi(sy, 0, H); r1[jx] = clipi(sy + 1, 0, H);
                }
                float4 tp[2][4];
#pragma unroll
                for (int jx = 0; jx < 2; jx++) {
                    if (!from_global) {
                        const int q0 = sx0[jx] - c0, q1 = sx1[jx] - c0;
                        tp[jx][0] = band[r0[jx] * BW + q0]; tp[jx][1] = band[r0[jx] * BW + q1];
                        tp[jx][2] = band[r1[jx] * BW + q0]; tp[jx][3] = band[r1[jx] * BW + q1];
                    } else {
                        const float3 g0 = att[(size_t)r0[jx] * W + sx0[jx]], g1 = att[(size_t)r0[jx] * W + sx1[jx]];
                        const float3 g2 = att[(size_t)r1[jx] * W + sx0[jx]], g3 = att[(size_t)r1[jx] * W + sx1[jx]];
                        tp[jx][0] = make_float4(g0.x, g0.y, g0.z, 0.f); tp[jx][1] = make_float4(g1.x, g1.y, g1.z, 0.f);
                        tp[jx][2] = make_float4(g2.x, g2.y, g2.z, 0.f); tp[jx][3] = make_float4(g3.x, g3.y, g3.z, 0.f);
                    }
                }
#pragma unroll
                for (int jx = 0; jx < 2; jx++) {
                    const float v0 = tp[jx][0].x * w[jx][0] + tp[jx][1].x * w[jx][1] + tp[jx][2].x * w[jx][2] + tp[jx][3].x * w[jx][3];
                    const float v1 = tp[jx][0].y * w[jx][0] + tp[jx][1].y * w[jx][1] + tp[jx][2].y * w[jx][2] + tp[jx][3].y * w[jx][3];
                    const float v2 = tp[jx][0].z * w[jx][0] + tp[jx][1].z * w[jx][1] + tp[jx][2].z * w[jx][2] + tp[jx][3].z * w[jx][3];
                    gq[jy][jx][0] = f255[lut[((int)(v0 * 255.0f)) & 255]];
                    gq[jy][jx][1] = f255[lut[((int)(v1 * 255.0f)) & 255]];
                    gq[jy][jx][2] = f255[lut[((int)(v2 * 255.0f)) & 255]];
                }
            }
            const float txw0 = im->txw0[ox], txw1 = im->txw1[ox];
            float v[3];
#pragma unroll
            for (int ch3 = 0; ch3 < 3; ch3++) {
                float r;
                if (same) r = gq[0][0][ch3];
                else if (area2) r = ((((0.f + gq[0][0][ch3]) + gq[0][1][ch3]) + gq[1][0][ch3]) + gq[1][1][ch3]) * 0.25f;
                else {
                    const float h0 = gq[0][0][ch3] * txw0 + gq[0][1][ch3] * txw1;
                    const float h1 = gq[1][0][ch3] * txw0 + gq[1][1][ch3] * txw1;
                    r = h0 * ty.w0 + h1 * ty.w1;
                }
                r = clamp01(r);
                v[ch3] = normalize ? r - 0.5f : r;
            }
            const int p = oy * 32 + ox;
            const size_t local = (size_t)im->local;
            if (OUT == GR_OUT_CHW) {
                float* o = out + local * 3 * net * net;
                for (int ch3 = 0; ch3 < 3; ch3++) o[ch3 * net * net + p] = v[ch3];
            } else {
                float4 o4;
                if (OUT == GR_OUT_NHWC4_C11) {
                    float r[3];
#pragma unroll
                    for (int q = 0; q < 3; q++) {
                        float a1 = 0.f;
                        a1 = fmaf(v[0], s_c11[q * 3 + 0], a1);
                        a1 = fmaf(v[1], s_c11[q * 3 + 1], a1);
                        a1 = fmaf(v[2], s_c11[q * 3 + 2], a1);
                        a1 = a1 + s_c11[9 + q];
                        r[q] = a1 > 0.f ? a1 : 0.f;
                    }
                    o4 = make_float4(r[0], r[1], r[2], 0.f);
                } else {
                    o4 = make_float4(v[0], v[1], v[2], 0.f);
                }
                reinterpret_cast<float4*>(out)[local * (net * net) + p] = o4;
            }
        }
        __syncthreads();                          // every wave is done sampling this band (and reading its list)
        if (tid == 0) *s_cnt = 0;
        if (b + 1 < nb) stash();
        __syncthreads();
    }
}

// stand-alone warpPerspective, planar src -> planar dst (no gamma)
__global__ void k_warp_full(const float* __restrict__ src, int H, int W, double m0, double m1, double m2, double m3,
                            double m4, double m5, double m6, double m7, double m8, int bw0, int cw, int ch,
                            float* __restrict__ dst) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= cw * ch) return;
    int y = idx / cw, x = idx % cw;
    double Mi[9] = {m0, m1, m2, m3, m4, m5, m6, m7, m8};
    WarpTaps t = warp_taps(Mi, bw0, x, y, H, W);
    size_t pl = (size_t)H * W, cpl = (size_t)cw * ch;
    for (int c = 0; c < 3; c++) {
        const float* p = src + c * pl;
        dst[c * cpl + idx] = p[t.o00] * t.w0 + p[t.o01] * t.w1 + p[t.o10] * t.w2 + p[t.o11] * t.w3;
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
static inline double inv_scale(int ssize, int dsize) { return 1.0 / ((double)dsize / (double)ssize); }

int px_resize(gr_ctx* ctx, const float* src, int C, int sh, int sw, float* dst, int dh, int dw, int clamp, int normalize,
              hipStream_t st) {
    int total = C * dh * dw;
    hipLaunchKernelGGL(k_resize, dim3((total + 255) / 256), dim3(256), 0, st, src, C, sh, sw, dst, dh, dw,
                       inv_scale(sw, dw), inv_scale(sh, dh), clamp, normalize);
    GR_HIP(ctx, hipGetLastError());
    return GR_OK;
}

int px_add_noise(gr_ctx* ctx, const float* img, int H, int W, const float* mask, const float* theta, int nh, int nw,
                 float lbd, int clip, float* comb, float* pert, float* mask_lg, hipStream_t st) {
    int total = 3 * H * W;
    hipLaunchKernelGGL(k_add_noise, dim3((total + 255) / 256), dim3(256), 0, st, img, H, W, mask, theta, nh, nw,
                       inv_scale(nw, W), inv_scale(nh, H), lbd, clip, comb, pert, mask_lg);
    GR_HIP(ctx, hipGetLastError());
    return GR_OK;
}

int px_warp_full(gr_ctx* ctx, const float* src, int H, int W, const double* Mi, int crop_x, int crop_y, float* dst,
                 hipStream_t st) {
    int bh0 = 16 < crop_y ? 16 : crop_y;
    int bw0 = 1024 / bh0;
    if (bw0 > crop_x) bw0 = crop_x;
    int total = crop_x * crop_y;
    hipLaunchKernelGGL(k_warp_full, dim3((total + 255) / 256), dim3(256), 0, st, src, H, W, Mi[0], Mi[1], Mi[2], Mi[3],
                       Mi[4], Mi[5], Mi[6], Mi[7], Mi[8], bw0, crop_x, crop_y, dst);
    GR_HIP(ctx, hipGetLastError());
    return GR_OK;
}

int px_composite(gr_ctx* ctx, const float* img, int H, int W, const float* masks, const float* thetas, int ncand,
                 int nh, int nw, float lbd, const gr_xforms* xf, float* att, hipStream_t st) {
    if (ncand <= 0) return GR_OK;
    int tiles = ((W + CT - 1) / CT) * ((H + CT - 1) / CT);
    // algorithmic bytes: image once + (mask, theta) per candidate + one RGB composite per (candidate, blur)
    ProfScope ps_(ctx, "k_composite", st, 0.0,
                  12.0 * H * W + (double)ncand * (24.0 * nh * nw + (double)ATT_B * xf->nslots * H * W));
    for (int c0 = 0; c0 < ncand; c0 += 32768) {        // gridDim.y limit
        int nc = ncand - c0 < 32768 ? ncand - c0 : 32768;
        hipLaunchKernelGGL(k_composite<false>, dim3(tiles, nc), dim3(256), 0, st, img, H, W,
                           masks + (size_t)c0 * 3 * nh * nw, thetas + (size_t)c0 * 3 * nh * nw, nh, nw,
                           inv_scale(nw, W), inv_scale(nh, H), lbd, xf->nslots, xf->slot_blur[0], xf->slot_blur[1],
                           xf->slot_blur[2], xf->slot_blur[3], att + (size_t)c0 * xf->nslots * H * W * ATT_C);
    }
    GR_HIP(ctx, hipGetLastError());
    return GR_OK;
}

int px_composite_from_pert(gr_ctx* ctx, const float* img, const float* mask_lg, const float* pert, int H, int W,
                           const gr_xforms* xf, float* att, hipStream_t st) {
    int tiles = ((W + CT - 1) / CT) * ((H + CT - 1) / CT);
    hipLaunchKernelGGL(k_composite<true>, dim3(tiles, 1), dim3(256), 0, st, img, H, W, mask_lg, pert, H, W, 1.0, 1.0,
                       1.0f, xf->nslots, xf->slot_blur[0], xf->slot_blur[1], xf->slot_blur[2], xf->slot_blur[3], att);
    GR_HIP(ctx, hipGetLastError());
    return GR_OK;
}

void px_net_size(int crop_x, int crop_y, int net_size, int* nx, int* ny) { net_size_rule(crop_x, crop_y, net_size, nx, ny); }

int px_warp_net(gr_ctx* ctx, const float* att, int H, int W, const gr_xforms* xf, int64_t n_begin, int count, int cand0,
                int normalize, int net, float* out, hipStream_t st, int out_mode, const float* c11, int64_t var_stride) {
    if (count <= 0) return GR_OK;
    if (var_stride > 0 && out_mode != GR_OUT_CHW) return gr_fail(ctx, GR_EINVAL, "px_warp_net: variable-size output is CHW only");
    if (out_mode != GR_OUT_CHW && net != 32) return gr_fail(ctx, GR_EINVAL, "px_warp_net: NHWC4 output needs net == 32");
    if (out_mode == GR_OUT_NHWC4_C11 && !c11) return gr_fail(ctx, GR_EINVAL, "px_warp_net: conv1_1 parameters missing");
    // algorithmic bytes: 12 B/px written per image + 336 B of transform descriptor read
    ProfScope ps_(ctx, "k_warp_net", st, 0.0, (double)count * (12.0 * net * net + 336.0));
    // LDS-band kernel (opt-in, GRAPHITE_WARP_BAND=1): bit-identical to the gather kernel but MEASURED SLOWER on MI355X
    // (1.56-1.91 ms against 0.95 ms per 32,000 images, profiles/r02_warp_kernels.md): its inner loop is 600 instructions
    // and ~50 LDS operations per output-pixel wavefront, which costs more than the gather kernel's 16 divergent loads.
    static int band_on = -1;
    if (band_on < 0) { const char* e = getenv("GRAPHITE_WARP_BAND"); band_on = (e && e[0] == '1') ? 1 : 0; }
    if (band_on && net == 32 && var_stride == 0 && xf->all_sep && count >= 8 * WARP_BAND_K && H <= 2048) {
        constexpr int HALO = 12;                 // widest footprint of an output column that a band must hold
        const size_t tail = WB_TAIL;
        int BW = (int)(((size_t)160 * 1024 - tail) / ((size_t)H * 16));
        if (BW > WB_NT * WB_PRE / H) BW = WB_NT * WB_PRE / H;
        if (!(BW & 1)) BW--;                     // odd row stride (in 16-byte slots): the rows of a column tap different banks
        int S, nb;
        if (BW >= W) { BW = W | 1; S = W; nb = 1; }
        else { S = BW - HALO; nb = S > 0 ? (W + S - 1) / S : 0; }
        if ((size_t)H * BW > (size_t)WB_NT * WB_PRE) S = 0;
        if (S >= 8 && nb <= 253) {               // band_of keeps a band index in one byte, 254 / 255 are sentinels
            const int cand_lo = (int)(n_begin / xf->T), cand_hi = (int)((n_begin + count - 1) / xf->T);
            const int ncand = cand_hi - cand_lo + 1;
            const size_t smem = (size_t)H * BW * 16 + tail;
            const long long grid = (long long)((ncand + 7) / 8) * 8 * xf->ngroups;
            if (grid <= 0x7fffffffLL) {
#define WBK(M)                                                                                                              \
                do {                                                                                                        \
                    auto kern = k_warp_band<M>;                                                                             \
                    static std::atomic<bool> attr_set[64];                                                                     \
                    int dev_ = 0;                                                                                           \
                    GR_HIP(ctx, hipGetDevice(&dev_));                                                                       \
                    if (dev_ >= 0 && dev_ < 64 && !attr_set[dev_]) {                                                        \
                        GR_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                \
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));           \
                        attr_set[dev_] = true;                                                                              \
                    }                                                                                                       \
                    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WB_NT), smem, st, att, H, W, xf->nslots, xf->dev,   \
                                       xf->order_dev, xf->groups_dev, xf->ngroups, xf->T, (long long)n_begin, count, cand_lo, \
                                       ncand, cand0, normalize, BW, S, nb, c11, out);                                       \
                } while (0)
                if (out_mode == GR_OUT_CHW) WBK(GR_OUT_CHW);
                else if (out_mode == GR_OUT_NHWC4) WBK(GR_OUT_NHWC4);
                else WBK(GR_OUT_NHWC4_C11);
#undef WBK
                GR_HIP(ctx, hipGetLastError());
                return GR_OK;
            }
        }
    }
    static int xcd_map = -1;                     // GRAPHITE_WARP_XCD=0 turns the XCD-aware block order off (A/B runs)
    if (xcd_map < 0) { const char* e = getenv("GRAPHITE_WARP_XCD"); xcd_map = (e && e[0] == '0') ? 0 : 1; }
    const int use_map = (xcd_map && count >= 64) ? 1 : 0;
    const int grid = use_map ? 8 * ((count + 7) / 8) : count;
    if (net != 32 || var_stride > 0) {
        hipLaunchKernelGGL(k_warp_net_any, dim3(grid), dim3(256), 0, st, att, H, W, xf->nslots, xf->dev, xf->T,
                           (long long)n_begin, cand0, normalize, net, var_stride > 0 ? 1 : 0,
                           (long long)(var_stride > 0 ? var_stride : (int64_t)3 * net * net), count, use_map, out);
    } else {
        static int quad = -1;                    // GRAPHITE_WARP_QUAD=0: the one-lane-per-output-pixel kernel (A/B runs)
        if (quad < 0) { const char* e = getenv("GRAPHITE_WARP_QUAD"); quad = (e && e[0] == '0') ? 0 : 1; }
#define WN(K, M) hipLaunchKernelGGL(K<M>, dim3(grid), dim3(256), 0, st, att, H, W, xf->nslots, xf->dev, xf->T, \
                                    (long long)n_begin, cand0, normalize, count, use_map, c11, out)
        if (quad && xf->all_sep && H <= 32768 && W <= (1 << 19) && (long long)H * W * ATT_B <= 0xffffffffLL) {   // row table packs a source row in 16 bits
            if (out_mode == GR_OUT_CHW) WN(k_warp_quad, GR_OUT_CHW);
            else if (out_mode == GR_OUT_NHWC4) WN(k_warp_quad, GR_OUT_NHWC4);
            else WN(k_warp_quad, GR_OUT_NHWC4_C11);
        } else {
            if (out_mode == GR_OUT_CHW) WN(k_warp_net32, GR_OUT_CHW);
            else if (out_mode == GR_OUT_NHWC4) WN(k_warp_net32, GR_OUT_NHWC4);
            else WN(k_warp_net32, GR_OUT_NHWC4_C11);
        }
#undef WN
    }
    GR_HIP(ctx, hipGetLastError());
    return GR_OK;
}
