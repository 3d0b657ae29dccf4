// pixel.hip -- the image half of the transform-robustness path, hand-written for gfx950.
//
//   k_composite   add_noise (transforms.py:143-159) + GaussianBlur/where/clip (transforms.py:55-62)
//                 fused per 32x32 tile; one RGBX f32 composite per (candidate, blur kernel).
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
        float* dst = att + ((size_t)cand * nslots + s) * pl * 4;
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
                *reinterpret_cast<float4*>(dst + ((size_t)gy * W + gx) * 4) = make_float4(o[0], o[1], o[2], 0.f);
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

// warped + gamma'd pixel from an RGBX composite
__device__ __forceinline__ void warp_gamma_rgbx(const float4* __restrict__ att, const double* Mi, int bw0, int x, int y,
                                                int H, int W, const float* lutf, float* o3) {
    WarpTaps t = warp_taps(Mi, bw0, x, y, H, W);
    float4 a = att[t.o00], b = att[t.o01], c = att[t.o10], d = att[t.o11];
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
    const float4* att = reinterpret_cast<const float4*>(att_all) + ((size_t)cand * nslots + s_par[2]) * pl;
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
    const float4* att = reinterpret_cast<const float4*>(att_all) + ((size_t)cand * nslots + x->slot) * pl;
    const double m4 = s_Mi[4], m5 = s_Mi[5];
    const int nq = same ? 1 : 2;
#pragma unroll 1
    for (int p = tid; p < net * net; p += 256) {
        const int oy = p >> 5, ox = p & 31;
        const Tap tx = s_tx[ox], ty = s_ty[oy];
        float g[2][2][3];
        for (int jy = 0; jy < nq; jy++) {
            const int y = jy ? ty.i1 : ty.i0;
            for (int jx = 0; jx < nq; jx++) {
                if (sep) {
                    const WarpCol c = s_col[ox * 2 + jx];
                    const double Y0 = (c.A + m4 * y) + m5;
                    const int Y = cv_round_sat((Y0 + c.B) * c.Wd);
                    const int sy = sat_short(Y >> 5);
                    const float fy = (float)(Y & 31) * (1.f / 32.f), fx = c.fx;
                    const float w0 = (1.f - fy) * (1.f - fx), w1 = (1.f - fy) * fx, w2 = fy * (1.f - fx), w3 = fy * fx;
                    const int r0 = clipi(sy, 0, H) * W, r1 = clipi(sy + 1, 0, H) * W;
                    const float4 a = att[r0 + c.sx0], b = att[r0 + c.sx1], cc = att[r1 + c.sx0], d = att[r1 + c.sx1];
                    g[jy][jx][0] = gamma_stage_f(a.x * w0 + b.x * w1 + cc.x * w2 + d.x * w3, s_lutf);
                    g[jy][jx][1] = gamma_stage_f(a.y * w0 + b.y * w1 + cc.y * w2 + d.y * w3, s_lutf);
                    g[jy][jx][2] = gamma_stage_f(a.z * w0 + b.z * w1 + cc.z * w2 + d.z * w3, s_lutf);
                } else {
                    warp_gamma_rgbx(att, s_Mi, bw0, jx ? tx.i1 : tx.i0, y, H, W, s_lutf, g[jy][jx]);
                }
            }
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
    // algorithmic bytes: image once + (mask, theta) per candidate + one RGBX composite per (candidate, blur)
    ProfScope ps_(ctx, "k_composite", st, 0.0,
                  12.0 * H * W + (double)ncand * (24.0 * nh * nw + 16.0 * xf->nslots * H * W));
    for (int c0 = 0; c0 < ncand; c0 += 32768) {        // gridDim.y limit
        int nc = ncand - c0 < 32768 ? ncand - c0 : 32768;
        hipLaunchKernelGGL(k_composite<false>, dim3(tiles, nc), dim3(256), 0, st, img, H, W,
                           masks + (size_t)c0 * 3 * nh * nw, thetas + (size_t)c0 * 3 * nh * nw, nh, nw,
                           inv_scale(nw, W), inv_scale(nh, H), lbd, xf->nslots, xf->slot_blur[0], xf->slot_blur[1],
                           xf->slot_blur[2], xf->slot_blur[3], att + (size_t)c0 * xf->nslots * H * W * 4);
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
    static int xcd_map = -1;                     // GRAPHITE_WARP_XCD=0 turns the XCD-aware block order off (A/B runs)
    if (xcd_map < 0) { const char* e = getenv("GRAPHITE_WARP_XCD"); xcd_map = (e && e[0] == '0') ? 0 : 1; }
    const int use_map = (xcd_map && count >= 64) ? 1 : 0;
    const int grid = use_map ? 8 * ((count + 7) / 8) : count;
    if (net != 32 || var_stride > 0) {
        hipLaunchKernelGGL(k_warp_net_any, dim3(grid), dim3(256), 0, st, att, H, W, xf->nslots, xf->dev, xf->T,
                           (long long)n_begin, cand0, normalize, net, var_stride > 0 ? 1 : 0,
                           (long long)(var_stride > 0 ? var_stride : (int64_t)3 * net * net), count, use_map, out);
    } else {
#define WN(M) hipLaunchKernelGGL(k_warp_net32<M>, dim3(grid), dim3(256), 0, st, att, H, W, xf->nslots, xf->dev, xf->T, \
                                 (long long)n_begin, cand0, normalize, count, use_map, c11, out)
        if (out_mode == GR_OUT_CHW) WN(GR_OUT_CHW);
        else if (out_mode == GR_OUT_NHWC4) WN(GR_OUT_NHWC4);
        else WN(GR_OUT_NHWC4_C11);
#undef WN
    }
    GR_HIP(ctx, hipGetLastError());
    return GR_OK;
}
