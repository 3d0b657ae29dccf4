// wb.hip -- the differentiable ("whitebox") twin of the transform, forward and backward, on gfx950.
//
// Reference: transforms.py:176-228 (transform_wb, convert2NetworkWB) with perspective_transform.py:49
// (kornia.warp_perspective(align_corners=True, padding_mode='border')), kornia.enhance.adjust_gamma and
// kornia resize(32, align_corners=False) = F.interpolate bilinear.  Restated in oracle/wb_reference.py (torch).
//
//   composite   att_c = clamp(att,0,1); blur k: where(mask, conv2d_zero_pad(where(mask, att_c, 0), outer(g,g)), orig)
//                                        k = 0: where(mask, att_c, orig);   clamp           -> one RGBX plane per blur kernel in use
//   sample      per (transform, output pixel): F.interpolate taps (<= 2x2 pixels of the crop_y x crop_x warped image),
//               each evaluated ON DEMAND: normalised dst coords -> A (3x3 f32) -> grid_sample bilinear, border clamp ->
//               clamp(pow(v, gamma), 0, 1); result clamped to [0,1]
//   backward    the same traversal scatters d out / d composite with atomicAdd (f32) into a gradient plane per blur
//               kernel; a second kernel takes it through the blur (self-adjoint, zero padded), the mask and the first clamp.
// All arithmetic is float32, as in the reference (the double-precision cv2 path is pixel.hip).
#include "common.h"
#include <math.h>

namespace {

struct WbDev { float A[9]; int crop_x, crop_y; float gamma; int slot; };

__device__ __forceinline__ float clamp01f(float v) { return fminf(fmaxf(v, 0.f), 1.f); }

__constant__ float c_gauss[3][7] = {{0.25f, 0.5f, 0.25f, 0.f, 0.f, 0.f, 0.f},
                                    {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f, 0.f, 0.f},
                                    {0.03125f, 0.109375f, 0.21875f, 0.28125f, 0.21875f, 0.109375f, 0.03125f}};
// k2[i][j] = (float)((double)g[i] * (double)g[j]): every g is a dyadic rational, so the float product is exact
__device__ __forceinline__ float k2(int k, int i, int j) { return c_gauss[(k - 3) >> 1][i] * c_gauss[(k - 3) >> 1][j]; }

// ---- composites: comp[slot][H][W][4] ----
__global__ void k_wb_composite(const float* __restrict__ orig, const float* __restrict__ att, const float* __restrict__ mask,
                               int H, int W, int nslots, int b0, int b1, int b2, int b3, float* __restrict__ comp) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= H * W) return;
    const int y = idx / W, x = idx % W;
    const size_t pl = (size_t)H * W;
    const int blurs[4] = {b0, b1, b2, b3};
    for (int s = 0; s < nslots; s++) {
        const int k = blurs[s], r = k >> 1;
        float o[3];
        for (int c = 0; c < 3; c++) {
            const bool m = mask[c * pl + idx] > 0.5f;
            float v;
            if (!m) {
                v = orig[c * pl + idx];
            } else if (k == 0) {
                v = clamp01f(att[c * pl + idx]);
            } else {
                v = 0.f;
                for (int dy = 0; dy < k; dy++) {
                    const int yy = y + dy - r;
                    if (yy < 0 || yy >= H) continue;
                    for (int dx = 0; dx < k; dx++) {
                        const int xx = x + dx - r;
                        if (xx < 0 || xx >= W) continue;
                        const size_t q = c * pl + (size_t)yy * W + xx;
                        if (mask[q] > 0.5f) v += k2(k, dy, dx) * clamp01f(att[q]);
                    }
                }
            }
            o[c] = clamp01f(v);
        }
        reinterpret_cast<float4*>(comp)[(size_t)s * pl + idx] = make_float4(o[0], o[1], o[2], 0.f);
    }
}

// ---- sampling geometry shared by forward and backward ----
struct Bil { int o[4]; float w[4]; };   // up to 4 in-bounds taps (offset -1 = out of bounds, weight unused)

// grid_sample(align_corners=True, padding_mode='border') taps of warped pixel (x, y) of a crop_y x crop_x image
__device__ __forceinline__ Bil wb_taps(const float* A, int x, int y, int cw, int ch, int H, int W) {
    const float xn = ((float)x / (float)(cw - 1) - 0.5f) * 2.f;
    const float yn = ((float)y / (float)(ch - 1) - 0.5f) * 2.f;
    const float q0 = A[0] * xn + A[1] * yn + A[2];
    const float q1 = A[3] * xn + A[4] * yn + A[5];
    const float z = A[6] * xn + A[7] * yn + A[8];
    const float sc = fabsf(z) > 1e-8f ? 1.0f / (z + 1e-8f) : 1.0f;
    const float gx = q0 * sc, gy = q1 * sc;
    float ix = ((gx + 1.f) / 2.f) * (float)(W - 1);
    float iy = ((gy + 1.f) / 2.f) * (float)(H - 1);
    ix = fminf(fmaxf(ix, 0.f), (float)(W - 1));
    iy = fminf(fmaxf(iy, 0.f), (float)(H - 1));
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    const int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
    const float tx = ix - fx0, ty = iy - fy0;
    Bil b;
    b.w[0] = (1.f - tx) * (1.f - ty); b.w[1] = tx * (1.f - ty); b.w[2] = (1.f - tx) * ty; b.w[3] = tx * ty;
    const bool xi0 = x0 >= 0 && x0 < W, xi1 = x1 >= 0 && x1 < W, yi0 = y0 >= 0 && y0 < H, yi1 = y1 >= 0 && y1 < H;
    b.o[0] = (xi0 && yi0) ? y0 * W + x0 : -1;
    b.o[1] = (xi1 && yi0) ? y0 * W + x1 : -1;
    b.o[2] = (xi0 && yi1) ? y1 * W + x0 : -1;
    b.o[3] = (xi1 && yi1) ? y1 * W + x1 : -1;
    return b;
}

struct Lin { int i0, i1; float l0, l1; };
// F.interpolate(mode='bilinear', align_corners=False) source taps of output index d (area_pixel_compute_source_index)
__device__ __forceinline__ Lin wb_lin(int d, int in, float scale) {
    float s = scale * ((float)d + 0.5f) - 0.5f;
    if (s < 0.f) s = 0.f;
    Lin t;
    t.i0 = (int)s;
    if (t.i0 > in - 1) t.i0 = in - 1;
    t.i1 = t.i0 + (t.i0 < in - 1 ? 1 : 0);
    t.l1 = s - (float)t.i0;
    t.l0 = 1.f - t.l1;
    return t;
}

__device__ __forceinline__ void wb_sample(const float4* __restrict__ comp, const Bil& b, float* v3) {
    v3[0] = v3[1] = v3[2] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (b.o[k] >= 0) {
            const float4 p = comp[b.o[k]];
            v3[0] += p.x * b.w[k]; v3[1] += p.y * b.w[k]; v3[2] += p.z * b.w[k];
        }
}

// One workgroup per transform; out[t][c][oy][ox].  BWD: gout is d loss / d out, gcomp the gradient planes.
template <bool BWD>
__global__ __launch_bounds__(256) void k_wb_warp(const float* __restrict__ comp_all, int H, int W, const WbDev* __restrict__ xf,
                                                 int net, float* __restrict__ out, const float* __restrict__ gout,
                                                 float* __restrict__ gcomp_all) {
    __shared__ WbDev s_x;
    const int t = blockIdx.x;
    if (threadIdx.x == 0) s_x = xf[t];
    __syncthreads();
    const int cw = s_x.crop_x, ch = s_x.crop_y;
    const float gamma = s_x.gamma;
    const size_t pl = (size_t)H * W;
    const float4* comp = reinterpret_cast<const float4*>(comp_all) + (size_t)s_x.slot * pl;
    float* gcomp = BWD ? gcomp_all + (size_t)s_x.slot * pl * 4 : nullptr;
    const float scx = (float)cw / (float)net, scy = (float)ch / (float)net;
    const int npx = net * net;
    for (int p = threadIdx.x; p < npx; p += 256) {
        const int oy = p / net, ox = p % net;
        const Lin lx = wb_lin(ox, cw, scx), ly = wb_lin(oy, ch, scy);
        const int xs[2] = {lx.i0, lx.i1}, ys[2] = {ly.i0, ly.i1};
        const float wx[2] = {lx.l0, lx.l1}, wy[2] = {ly.l0, ly.l1};
        if (!BWD) {
            float G[2][2][3];
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 2; b++) {
                    const Bil tp = wb_taps(s_x.A, xs[b], ys[a], cw, ch, H, W);
                    float v[3];
                    wb_sample(comp, tp, v);
#pragma unroll
                    for (int c = 0; c < 3; c++) G[a][b][c] = clamp01f(powf(v[c], gamma));
                }
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float r = wy[0] * (wx[0] * G[0][0][c] + wx[1] * G[0][1][c]) + wy[1] * (wx[0] * G[1][0][c] + wx[1] * G[1][1][c]);
                out[((size_t)t * 3 + c) * npx + p] = clamp01f(r);
            }
        } else {
            float go[3];
#pragma unroll
            for (int c = 0; c < 3; c++) go[c] = gout[((size_t)t * 3 + c) * npx + p];
            // (the outer clamp and adjust_gamma's clamp always pass: their arguments are convex mixes of values in [0,1])
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 2; b++) {
                    const float wr = wy[a] * wx[b];
                    const Bil tp = wb_taps(s_x.A, xs[b], ys[a], cw, ch, H, W);
                    float v[3];
                    wb_sample(comp, tp, v);
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        const float gv = (go[c] * wr) * (gamma * powf(v[c], gamma - 1.f));     // d clamp(pow(v, g)) / d v
#pragma unroll
                        for (int k = 0; k < 4; k++)
                            if (tp.o[k] >= 0) atomicAdd(gcomp + (size_t)tp.o[k] * 4 + c, gv * tp.w[k]);
                    }
                }
        }
    }
}

// gradient planes -> d loss / d att  (blur is self-adjoint: symmetric kernel, zero padding)
__global__ void k_wb_composite_bwd(const float* __restrict__ att, const float* __restrict__ mask, int H, int W, int nslots,
                                   int b0, int b1, int b2, int b3, const float* __restrict__ gcomp, float* __restrict__ gatt) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= H * W) return;
    const int y = idx / W, x = idx % W;
    const size_t pl = (size_t)H * W;
    const int blurs[4] = {b0, b1, b2, b3};
    for (int c = 0; c < 3; c++) {
        float g = 0.f;
        const float a = att[c * pl + idx];
        if (mask[c * pl + idx] > 0.5f && a >= 0.f && a <= 1.f) {
            for (int s = 0; s < nslots; s++) {
                const int k = blurs[s], r = k >> 1;
                const float* gs = gcomp + (size_t)s * pl * 4;
                if (k == 0) {
                    g += gs[(size_t)idx * 4 + c];
                } else {
                    for (int dy = 0; dy < k; dy++) {
                        const int yy = y + dy - r;
                        if (yy < 0 || yy >= H) continue;
                        for (int dx = 0; dx < k; dx++) {
                            const int xx = x + dx - r;
                            if (xx < 0 || xx >= W) continue;
                            const size_t q = (size_t)yy * W + xx;
                            if (mask[c * pl + q] > 0.5f) g += k2(k, dy, dx) * gs[q * 4 + c];
                        }
                    }
                }
            }
        }
        gatt[c * pl + idx] = g;
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// entry points
// ---------------------------------------------------------------------------------------------
static int wb_prepare(gr_ctx* ctx, const float* orig, const float* att, const float* mask, int H, int W, const gr_wb_xform* xf,
                      int T, int net, hipStream_t st, int* nslots_out, int* blurs, float** comp, WbDev** dev, float** gcomp,
                      bool want_grad) {
    if (!orig || !att || !mask || !xf || H <= 1 || W <= 1 || T <= 0 || net <= 0 || net > 1024)
        return gr_fail(ctx, GR_EINVAL, "gr_wb_*: bad arguments");
    if ((long long)H * W > (1 << 26)) return gr_fail(ctx, GR_EINVAL, "gr_wb_*: frame too large");
    int slot_of[8], nslots = 0;
    for (int i = 0; i < 8; i++) slot_of[i] = -1;
    for (int b = 0; b <= 7; b++) {
        bool used = false;
        for (int t = 0; t < T && !used; t++) used = xf[t].blur == b;
        if (used) {
            if (!(b == 0 || b == 3 || b == 5 || b == 7)) return gr_fail(ctx, GR_EINVAL, "unsupported blur kernel %d", b);
            blurs[nslots] = b;
            slot_of[b] = nslots++;
        }
    }
    std::vector<WbDev> h(T);
    for (int t = 0; t < T; t++) {
        if (xf[t].blur < 0 || xf[t].blur > 7 || slot_of[xf[t].blur] < 0) return gr_fail(ctx, GR_EINVAL, "unsupported blur kernel %d", xf[t].blur);
        if (xf[t].crop_x <= 1 || xf[t].crop_y <= 1 || xf[t].crop_x > 16384 || xf[t].crop_y > 16384)
            return gr_fail(ctx, GR_EINVAL, "transform %d: bad crop %dx%d", t, xf[t].crop_x, xf[t].crop_y);
        for (int i = 0; i < 9; i++) h[t].A[i] = xf[t].A[i];
        h[t].crop_x = xf[t].crop_x; h[t].crop_y = xf[t].crop_y; h[t].gamma = xf[t].gamma; h[t].slot = slot_of[xf[t].blur];
    }
    const size_t plane = gr_align((size_t)nslots * H * W * 16);
    const size_t tab = gr_align(sizeof(WbDev) * (size_t)T);
    int rc = gr_ws_reserve(ctx, plane * (want_grad ? 2 : 1) + tab, st);
    if (rc) return rc;
    *comp = (float*)ctx->ws;
    *gcomp = want_grad ? (float*)(ctx->ws + plane) : nullptr;
    *dev = (WbDev*)(ctx->ws + plane * (want_grad ? 2 : 1));
    // pageable host source: the copy is staged before the call returns, so the vector may die with this frame
    GR_HIP(ctx, hipMemcpyAsync(*dev, h.data(), sizeof(WbDev) * (size_t)T, hipMemcpyHostToDevice, st));
    GR_HIP(ctx, hipStreamSynchronize(st));
    for (int i = nslots; i < 4; i++) blurs[i] = 0;
    *nslots_out = nslots;
    const int px = H * W;
    ProfScope ps_(ctx, "k_wb_composite", st, 0.0, 36.0 * px + 16.0 * nslots * px);
    hipLaunchKernelGGL(k_wb_composite, dim3((px + 255) / 256), dim3(256), 0, st, orig, att, mask, H, W, nslots, blurs[0],
                       blurs[1], blurs[2], blurs[3], *comp);
    GR_HIP(ctx, hipGetLastError());
    return GR_OK;
}

extern "C" int gr_wb_forward(gr_ctx* ctx, const float* orig, const float* att, const float* mask, int H, int W,
                             const gr_wb_xform* xf, int T, int net, float* out, void* stream) {
    if (!ctx) return GR_EINVAL;
    GR_HIP(ctx, hipSetDevice(ctx->device));
    if (!out) return gr_fail(ctx, GR_EINVAL, "gr_wb_forward: out is null");
    hipStream_t st = (hipStream_t)stream;
    int nslots = 0, blurs[4];
    float *comp = nullptr, *gcomp = nullptr;
    WbDev* dev = nullptr;
    int rc = wb_prepare(ctx, orig, att, mask, H, W, xf, T, net, st, &nslots, blurs, &comp, &dev, &gcomp, false);
    if (rc) return rc;
    ProfScope ps_(ctx, "k_wb_warp<fwd>", st, 0.0, (double)T * (12.0 * net * net + sizeof(WbDev)));
    hipLaunchKernelGGL(k_wb_warp<false>, dim3(T), dim3(256), 0, st, comp, H, W, dev, net, out, (const float*)nullptr,
                       (float*)nullptr);
    GR_HIP(ctx, hipGetLastError());
    return GR_OK;
}

extern "C" int gr_wb_backward(gr_ctx* ctx, const float* orig, const float* att, const float* mask, int H, int W,
                              const gr_wb_xform* xf, int T, int net, const float* grad_out, float* grad_att, void* stream) {
    if (!ctx) return GR_EINVAL;
    GR_HIP(ctx, hipSetDevice(ctx->device));
    if (!grad_out || !grad_att) return gr_fail(ctx, GR_EINVAL, "gr_wb_backward: null gradient pointer");
    hipStream_t st = (hipStream_t)stream;
    int nslots = 0, blurs[4];
    float *comp = nullptr, *gcomp = nullptr;
    WbDev* dev = nullptr;
    int rc = wb_prepare(ctx, orig, att, mask, H, W, xf, T, net, st, &nslots, blurs, &comp, &dev, &gcomp, true);
    if (rc) return rc;
    const int px = H * W;
    GR_HIP(ctx, hipMemsetAsync(gcomp, 0, (size_t)nslots * px * 16, st));
    {
        ProfScope ps_(ctx, "k_wb_warp<bwd>", st, 0.0, (double)T * (12.0 * net * net + sizeof(WbDev)));
        hipLaunchKernelGGL(k_wb_warp<true>, dim3(T), dim3(256), 0, st, comp, H, W, dev, net, (float*)nullptr, grad_out, gcomp);
        GR_HIP(ctx, hipGetLastError());
    }
    ProfScope ps_(ctx, "k_wb_composite_bwd", st, 0.0, 36.0 * px + 16.0 * nslots * px);
    hipLaunchKernelGGL(k_wb_composite_bwd, dim3((px + 255) / 256), dim3(256), 0, st, att, mask, H, W, nslots, blurs[0], blurs[1],
                       blurs[2], blurs[3], gcomp, grad_att);
    GR_HIP(ctx, hipGetLastError());
    return GR_OK;
}
