// api.hip -- the extern "C" surface of libgraphite_hip.so (see include/graphite_hip.h).
#include "common.h"
#include <limits.h>
#include <stdarg.h>
#include <string.h>

static thread_local std::string g_create_err;

// Scratch of gr_transform_composited_static (the entry point that allocates nothing): the RGB composites (12-byte texels) of ONE
// (img, mask_lg, pert) triple, one per blur kernel in use (<= 4), for frames of at most GR_STATIC_MAX_PIXELS pixels.
// Every GPU the library runs on holds its own copy until the process ends.  In host code the NAME is only a handle for
// hipGetSymbolAddress (gr_ctx_create resolves it once per device, with that device current); kernels, memsets and copies
// take the resolved per-device pointer gr_ctx::static_scratch.
static __device__ __align__(256) unsigned char scratch[(size_t)4 * GR_STATIC_MAX_PIXELS * ATT_B];

int gr_fail(gr_ctx* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf; else g_create_err = buf;
    return code;
}

// Grow-only device workspace.  Growing synchronises the device (frees the old block).
int gr_ws_reserve(gr_ctx* ctx, size_t bytes, void* stream) {
    (void)stream;
    if (bytes <= ctx->ws_bytes) return GR_OK;
    if (ctx->ws) {
        GR_HIP(ctx, hipDeviceSynchronize());
        GR_HIP(ctx, hipFree(ctx->ws));
        ctx->ws = nullptr;
        ctx->ws_bytes = 0;
    }
    size_t want = gr_align(bytes + bytes / 8, 1 << 20);
    hipError_t e = hipMalloc((void**)&ctx->ws, want);
    if (e != hipSuccess) {
        ctx->ws = nullptr;
        return gr_fail(ctx, GR_ENOMEM, "workspace hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    }
    ctx->ws_bytes = want;
    return GR_OK;
}

// ---------------------------------------------------------------------------------------------
// per-kernel profiling (HIP events on the caller's stream)
// ---------------------------------------------------------------------------------------------
static void prof_drain(gr_ctx* c, size_t count);
static hipEvent_t prof_event(gr_ctx* c) {
    if (!c->prof_pool.empty()) { hipEvent_t e = c->prof_pool.back(); c->prof_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
ProfScope::ProfScope(gr_ctx* c, const char* name, hipStream_t s, double flops, double bytes) : ctx(c), st(s) {
    if (!c || !c->prof) return;
    for (size_t i = 0; i < c->prof_entries.size(); i++)
        if (c->prof_entries[i].name == name) { entry = (int)i; break; }
    if (entry < 0) { c->prof_entries.push_back(ProfEntry()); entry = (int)c->prof_entries.size() - 1; c->prof_entries[entry].name = name; }
    c->prof_entries[entry].launches++;
    c->prof_entries[entry].flops += flops;
    c->prof_entries[entry].bytes += bytes;
    // a long job (config 4: ~1,500 launches per step) must not hold an event pair per launch until the end: read back the
    // older half, whose kernels finished long ago, while >= 1024 launches are still queued behind them
    if (c->prof_pending.size() >= 2048) prof_drain(c, 1024);
    e0 = prof_event(c); e1 = prof_event(c);
    if (e0 && e1) (void)hipEventRecord(e0, st);
}
ProfScope::~ProfScope() {
    if (entry < 0 || !e0 || !e1) return;
    (void)hipEventRecord(e1, st);
    ctx->prof_pending.push_back({entry, e0, e1});
}
// Reads back the first `count` pending event pairs (all of them by default) and returns their events to the pool.
static void prof_drain(gr_ctx* c, size_t count) {
    if (count > c->prof_pending.size()) count = c->prof_pending.size();
    for (size_t i = 0; i < count; i++) {
        ProfPending& p = c->prof_pending[i];
        float ms = 0.f;
        if (hipEventSynchronize(p.e1) == hipSuccess && hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess)
            c->prof_entries[p.entry].ms += ms;
        c->prof_pool.push_back(p.e0);
        c->prof_pool.push_back(p.e1);
    }
    c->prof_pending.erase(c->prof_pending.begin(), c->prof_pending.begin() + count);
}
extern "C" int gr_profile_enable(gr_ctx* ctx, int on) {
    if (!ctx) return GR_EINVAL;
    if (!on) prof_drain(ctx, (size_t)-1);
    ctx->prof = on != 0;
    return GR_OK;
}
extern "C" int gr_profile_reset(gr_ctx* ctx) {
    if (!ctx) return GR_EINVAL;
    prof_drain(ctx, (size_t)-1);
    ctx->prof_entries.clear();
    return GR_OK;
}
extern "C" int gr_profile_read(gr_ctx* ctx, int max_entries, gr_prof_entry* out) {
    if (!ctx) return GR_EINVAL;
    prof_drain(ctx, (size_t)-1);
    int n = (int)ctx->prof_entries.size();
    if (!out) return n;
    if (n > max_entries) n = max_entries;
    for (int i = 0; i < n; i++) {
        const ProfEntry& e = ctx->prof_entries[i];
        memset(&out[i], 0, sizeof(out[i]));
        strncpy(out[i].name, e.name.c_str(), sizeof(out[i].name) - 1);
        out[i].launches = e.launches; out[i].ms = e.ms; out[i].flops = e.flops; out[i].bytes = e.bytes;
    }
    return n;
}

extern "C" int gr_ctx_create(int device, gr_ctx** out) {
    if (!out) return gr_fail(nullptr, GR_EINVAL, "gr_ctx_create: out is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return gr_fail(nullptr, GR_EHIP, "no HIP device: %s", hipGetErrorString(e));
    if (device < 0 || device >= n) return gr_fail(nullptr, GR_EINVAL, "device %d out of range (0..%d)", device, n - 1);
    e = hipSetDevice(device);
    if (e != hipSuccess) return gr_fail(nullptr, GR_EHIP, "hipSetDevice: %s", hipGetErrorString(e));
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return gr_fail(nullptr, GR_EHIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return gr_fail(nullptr, GR_ESTATE, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    void* sp = nullptr;
    e = hipGetSymbolAddress(&sp, HIP_SYMBOL(scratch));          // this device's copy (hipSetDevice(device) above)
    if (e != hipSuccess || !sp) return gr_fail(nullptr, GR_EHIP, "hipGetSymbolAddress(scratch): %s", hipGetErrorString(e));
    gr_ctx* c = new gr_ctx();
    c->device = device;
    c->static_scratch = (char*)sp;
    *out = c;
    return GR_OK;
}

extern "C" void gr_ctx_destroy(gr_ctx* ctx) {
    if (!ctx) return;
    if (ctx->ws) { (void)hipDeviceSynchronize(); (void)hipFree(ctx->ws); }
    prof_drain(ctx, (size_t)-1);
    for (hipEvent_t e : ctx->prof_pool) (void)hipEventDestroy(e);
    delete ctx;
}

extern "C" const char* gr_last_error(const gr_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

extern "C" int gr_ctx_set_chunk(gr_ctx* ctx, int n) {
    if (!ctx) return GR_EINVAL;
    if (n < 64 || n > (1 << 20)) return gr_fail(ctx, GR_EINVAL, "chunk must be in [64, 1048576]");
    ctx->chunk = n;
    return GR_OK;
}
extern "C" int64_t gr_ctx_workspace_bytes(const gr_ctx* ctx) { return ctx ? (int64_t)ctx->ws_bytes : 0; }
extern "C" int gr_ctx_set_workspace_limit(gr_ctx* ctx, int64_t bytes) {
    if (!ctx) return GR_EINVAL;
    if (bytes < ((int64_t)64 << 20)) return gr_fail(ctx, GR_EINVAL, "workspace limit must be at least 64 MiB");
    ctx->ws_limit = (size_t)bytes;
    return GR_OK;
}
// Images per classifier chunk for this model: the context's chunk, capped so that the classifier's scratch stays under
// the workspace limit (GTSRBNet keeps 176 KB per image, WideResNet-34-10 3.3 MB: 8192 images of it would be 27 GB).
static int chunk_for(const gr_ctx* ctx, const gr_model* m, int64_t want) {
    const size_t per = net_workspace_bytes(m, 1024) / 1024 + 1;
    int64_t cap = (int64_t)(ctx->ws_limit / per);
    if (cap < 64) cap = 64;
    int64_t c = ctx->chunk;
    if (c > cap) c = cap;
    if (c > want) c = want;
    return (int)c;
}

// ---------------------------------------------------------------------------------------------
// transform table
// ---------------------------------------------------------------------------------------------
// cv::invert for a 3x3 CV_64F matrix (closed-form adjugate; what cv::warpPerspective applies to M)
static void invert3x3(const double* S, double* t) {
    double det = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) + S[2] * (S[3] * S[7] - S[4] * S[6]);
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
static int warp_block_w(int width, int height) {     // column-block width of cv::WarpPerspectiveInvoker
    int bh0 = 16 < height ? 16 : height;
    int bw0 = 1024 / bh0;
    return bw0 > width ? width : bw0;
}

extern "C" int gr_xforms_create(gr_ctx* ctx, const gr_xform* h, int T, gr_xforms** out) {
    if (!ctx) return GR_EINVAL;
    if (!h || !out || T <= 0) return gr_fail(ctx, GR_EINVAL, "gr_xforms_create: bad arguments");
    GR_HIP(ctx, hipSetDevice(ctx->device));
    int slot_of[8];
    for (int i = 0; i < 8; i++) slot_of[i] = -1;
    gr_xforms* x = new gr_xforms();
    x->ctx = ctx; x->T = T;
    std::vector<XformDev> d(T);
    for (int b = 0; b <= 7; b++) {                   // slots in ascending kernel size
        bool used = false;
        for (int t = 0; t < T && !used; t++) used = h[t].blur == b;
        if (used) {
            if (!(b == 0 || b == 3 || b == 5 || b == 7)) { delete x; return gr_fail(ctx, GR_EINVAL, "unsupported blur kernel %d", b); }
            x->slot_blur[x->nslots] = b;
            slot_of[b] = x->nslots++;
        }
    }
    for (int t = 0; t < T; t++) {
        if (h[t].blur < 0 || h[t].blur > 7 || slot_of[h[t].blur] < 0) { delete x; return gr_fail(ctx, GR_EINVAL, "unsupported blur kernel %d", h[t].blur); }
        if (h[t].crop_x <= 0 || h[t].crop_y <= 0 || h[t].crop_x > 16384 || h[t].crop_y > 16384) {
            delete x;
            return gr_fail(ctx, GR_EINVAL, "transform %d: bad crop %dx%d", t, h[t].crop_x, h[t].crop_y);
        }
        invert3x3(h[t].M, d[t].Mi);
        d[t].crop_x = h[t].crop_x; d[t].crop_y = h[t].crop_y;
        x->crop.push_back({h[t].crop_x, h[t].crop_y});
        d[t].slot = slot_of[h[t].blur];
        d[t].bw0 = warp_block_w(h[t].crop_x, h[t].crop_y);
        d[t].sep = (d[t].Mi[1] == 0.0 && d[t].Mi[7] == 0.0) ? 1 : 0;
        d[t].pad_ = 0;
        memcpy(d[t].lut, h[t].lut, 256);
    }
    // slot-major order and workgroup groups for the LDS-band warp kernel
    std::vector<int> order;
    std::vector<WarpGroup> groups;
    x->all_sep = true;
    for (int t = 0; t < T; t++) x->all_sep = x->all_sep && d[t].sep;
    for (int s = 0; s < x->nslots; s++) {
        const int first = (int)order.size();
        for (int t = 0; t < T; t++)
            if (d[t].slot == s) order.push_back(t);
        for (int o = first; o < (int)order.size(); o += WARP_BAND_K) {
            const int n = (int)order.size() - o < WARP_BAND_K ? (int)order.size() - o : WARP_BAND_K;
            groups.push_back(WarpGroup{s, o, n, 0});
        }
    }
    x->ngroups = (int)groups.size();
    hipError_t e = hipMalloc((void**)&x->dev, sizeof(XformDev) * T);
    if (e == hipSuccess) e = hipMalloc((void**)&x->order_dev, sizeof(int) * T);
    if (e == hipSuccess) e = hipMalloc((void**)&x->groups_dev, sizeof(WarpGroup) * groups.size());
    if (e != hipSuccess) { gr_xforms_destroy(x); return gr_fail(ctx, GR_ENOMEM, "xforms hipMalloc: %s", hipGetErrorString(e)); }
    e = hipMemcpy(x->dev, d.data(), sizeof(XformDev) * T, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(x->order_dev, order.data(), sizeof(int) * T, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(x->groups_dev, groups.data(), sizeof(WarpGroup) * groups.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) { gr_xforms_destroy(x); return gr_fail(ctx, GR_EHIP, "xforms upload: %s", hipGetErrorString(e)); }
    *out = x;
    return GR_OK;
}
extern "C" void gr_xforms_destroy(gr_xforms* x) {
    if (!x) return;
    if (x->dev) (void)hipFree(x->dev);
    if (x->order_dev) (void)hipFree(x->order_dev);
    if (x->groups_dev) (void)hipFree(x->groups_dev);
    delete x;
}

// ---------------------------------------------------------------------------------------------
// pixel operators
// ---------------------------------------------------------------------------------------------
// every entry point makes the context's device current for the calling thread (kernel launches and the workspace
// allocation must not land on whatever device the caller last selected)
#define NEED_CTX() do { if (!ctx) return GR_EINVAL; GR_HIP(ctx, hipSetDevice(ctx->device)); } while (0)

extern "C" int gr_resize_linear(gr_ctx* ctx, const float* src, int C, int sh, int sw, float* dst, int dh, int dw, void* stream) {
    NEED_CTX();
    if (!src || !dst || C <= 0 || sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0) return gr_fail(ctx, GR_EINVAL, "gr_resize_linear: bad arguments");
    return px_resize(ctx, src, C, sh, sw, dst, dh, dw, 0, 0, (hipStream_t)stream);
}
extern "C" int gr_convert_to_network(gr_ctx* ctx, const float* src, int sh, int sw, float* dst, int net, int normalize, void* stream) {
    NEED_CTX();
    if (!src || !dst || sh <= 0 || sw <= 0 || net <= 0) return gr_fail(ctx, GR_EINVAL, "gr_convert_to_network: bad arguments");
    if (net != 32) {                                   // transforms.py:31-45: short side -> net, aspect kept, clamp only
        int nx, ny;
        px_net_size(sw, sh, net, &nx, &ny);
        return px_resize(ctx, src, 3, sh, sw, dst, ny, nx, 1, 0, (hipStream_t)stream);
    }
    return px_resize(ctx, src, 3, sh, sw, dst, net, net, 1, normalize, (hipStream_t)stream);
}
extern "C" int gr_add_noise(gr_ctx* ctx, const float* img, int H, int W, const float* mask, const float* theta, int nh, int nw,
                            float lbd, int clip, float* comb, float* pert, float* mask_lg, void* stream) {
    NEED_CTX();
    if (!img || !mask || !theta || H <= 0 || W <= 0 || nh <= 0 || nw <= 0) return gr_fail(ctx, GR_EINVAL, "gr_add_noise: bad arguments");
    return px_add_noise(ctx, img, H, W, mask, theta, nh, nw, lbd, clip, comb, pert, mask_lg, (hipStream_t)stream);
}
extern "C" int gr_warp_perspective(gr_ctx* ctx, const float* src, int H, int W, const double* M, int crop_x, int crop_y,
                                   float* dst, void* stream) {
    NEED_CTX();
    if (!src || !M || !dst || H <= 0 || W <= 0 || crop_x <= 0 || crop_y <= 0) return gr_fail(ctx, GR_EINVAL, "gr_warp_perspective: bad arguments");
    double Mi[9];
    invert3x3(M, Mi);
    return px_warp_full(ctx, src, H, W, Mi, crop_x, crop_y, dst, (hipStream_t)stream);
}

static int check_frame(gr_ctx* ctx, int H, int W, int nh, int nw, int net, const gr_xforms* xf) {
    if (!xf || xf->ctx != ctx) return gr_fail(ctx, GR_ESTATE, "transform table belongs to another context");
    if (H <= 0 || W <= 0 || nh <= 0 || nw <= 0 || net <= 0 || net > 1024) return gr_fail(ctx, GR_EINVAL, "bad frame sizes");
    if ((long long)H * W > (1 << 28)) return gr_fail(ctx, GR_EINVAL, "frame too large");
    return GR_OK;
}

extern "C" int gr_transform_composited(gr_ctx* ctx, const float* img, const float* mask_lg, const float* pert, int H, int W,
                                       const gr_xforms* xf, int normalize, int net, float* out, void* stream) {
    NEED_CTX();
    if (!img || !mask_lg || !pert || !out) return gr_fail(ctx, GR_EINVAL, "gr_transform_composited: null pointer");
    int rc = check_frame(ctx, H, W, H, W, net, xf);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    size_t att_bytes = gr_align((size_t)xf->nslots * H * W * ATT_B);
    if ((rc = gr_ws_reserve(ctx, att_bytes, stream))) return rc;
    float* att = (float*)ctx->ws;
    if ((rc = px_composite_from_pert(ctx, img, mask_lg, pert, H, W, xf, att, st))) return rc;
    return px_warp_net(ctx, att, H, W, xf, 0, xf->T, 0, normalize, net, out, st);
}

extern "C" int gr_net_size(int crop_x, int crop_y, int net_size, int* nx, int* ny) {
    if (!nx || !ny || crop_x <= 0 || crop_y <= 0 || net_size <= 0) return GR_EINVAL;
    if (net_size == 32) { *nx = *ny = 32; return GR_OK; }          // transforms.py:19-29: always 32 x 32
    px_net_size(crop_x, crop_y, net_size, nx, ny);
    return GR_OK;
}

// largest [3, ny, nx] image any transform of the table produces at this net_size, in floats
static int64_t var_image_floats(const gr_xforms* xf, int net_size) {
    int64_t mx = 0;
    for (int t = 0; t < xf->T; t++) {
        int nx, ny;
        px_net_size(xf->crop[t][0], xf->crop[t][1], net_size, &nx, &ny);
        if ((int64_t)3 * nx * ny > mx) mx = (int64_t)3 * nx * ny;
    }
    return mx;
}

extern "C" int gr_transform_composited_var(gr_ctx* ctx, const float* img, const float* mask_lg, const float* pert, int H, int W,
                                           const gr_xforms* xf, int net_size, float* out, int64_t out_stride, void* stream) {
    NEED_CTX();
    if (!img || !mask_lg || !pert || !out) return gr_fail(ctx, GR_EINVAL, "gr_transform_composited_var: null pointer");
    int rc = check_frame(ctx, H, W, H, W, net_size, xf);
    if (rc) return rc;
    if (net_size == 32) return gr_fail(ctx, GR_EINVAL, "net_size 32 is the fixed-size branch: use gr_transform_composited");
    if (out_stride < var_image_floats(xf, net_size)) return gr_fail(ctx, GR_EINVAL, "gr_transform_composited_var: out_stride too small");
    hipStream_t st = (hipStream_t)stream;
    size_t att_bytes = gr_align((size_t)xf->nslots * H * W * ATT_B);
    if ((rc = gr_ws_reserve(ctx, att_bytes, stream))) return rc;
    float* att = (float*)ctx->ws;
    if ((rc = px_composite_from_pert(ctx, img, mask_lg, pert, H, W, xf, att, st))) return rc;
    return px_warp_net(ctx, att, H, W, xf, 0, xf->T, 0, 0, net_size, out, st, GR_OUT_CHW, nullptr, out_stride);
}

extern "C" int gr_warp_batch_var(gr_ctx* ctx, const float* img, int H, int W, const float* masks, const float* thetas, int C,
                                 int nh, int nw, const gr_xforms* xf, float lbd, int net_size, float* out, int64_t out_stride,
                                 void* stream) {
    NEED_CTX();
    if (C == 0) return GR_OK;
    if (!img || !masks || !thetas || !out || C < 0) return gr_fail(ctx, GR_EINVAL, "gr_warp_batch_var: bad arguments");
    int rc = check_frame(ctx, H, W, nh, nw, net_size, xf);
    if (rc) return rc;
    if (net_size == 32) return gr_fail(ctx, GR_EINVAL, "net_size 32 is the fixed-size branch: use gr_warp_batch");
    if (out_stride < var_image_floats(xf, net_size)) return gr_fail(ctx, GR_EINVAL, "gr_warp_batch_var: out_stride too small");
    hipStream_t st = (hipStream_t)stream;
    const size_t per_cand = (size_t)xf->nslots * H * W * ATT_B;
    int cpass = (int)(((size_t)2 << 30) / per_cand);
    if (cpass < 1) cpass = 1;
    if (cpass > C) cpass = C;
    if ((rc = gr_ws_reserve(ctx, gr_align(per_cand * cpass), stream))) return rc;
    float* att = (float*)ctx->ws;
    const int T = xf->T;
    for (int c0 = 0; c0 < C; c0 += cpass) {
        int nc = C - c0 < cpass ? C - c0 : cpass;
        if ((rc = px_composite(ctx, img, H, W, masks + (size_t)c0 * 3 * nh * nw, thetas + (size_t)c0 * 3 * nh * nw, nc, nh, nw,
                               lbd, xf, att, st))) return rc;
        long long nimg = (long long)nc * T;
        for (long long o = 0; o < nimg; o += (1 << 30)) {
            int cnt = (int)(nimg - o < (1 << 30) ? nimg - o : (1 << 30));
            if ((rc = px_warp_net(ctx, att, H, W, xf, (int64_t)c0 * T + o, cnt, c0, 0, net_size,
                                  out + ((size_t)c0 * T + o) * (size_t)out_stride, st, GR_OUT_CHW, nullptr, out_stride))) return rc;
        }
    }
    return GR_OK;
}

// gr_transform_composited without any allocation: the composites live in this device's copy of `scratch`.
extern "C" int gr_transform_composited_static(gr_ctx* ctx, const float* img, const float* mask_lg, const float* pert, int H,
                                              int W, const gr_xforms* xf, int normalize, int net, float* out, void* stream) {
    NEED_CTX();
    if (!img || !mask_lg || !pert || !out) return gr_fail(ctx, GR_EINVAL, "gr_transform_composited_static: null pointer");
    int rc = check_frame(ctx, H, W, H, W, net, xf);
    if (rc) return rc;
    if ((long long)H * W > GR_STATIC_MAX_PIXELS)
        return gr_fail(ctx, GR_EINVAL, "gr_transform_composited_static: frame %dx%d exceeds the static scratch (at most %d pixels)",
                       H, W, GR_STATIC_MAX_PIXELS);
    if (ctx->prof) return gr_fail(ctx, GR_ESTATE, "gr_transform_composited_static: profiling creates HIP events; turn it off");
    if (!ctx->static_scratch) return gr_fail(ctx, GR_ESTATE, "static scratch not resolved");
    hipStream_t st = (hipStream_t)stream;
    float* att = (float*)ctx->static_scratch;             // every composite texel the warp reads is rewritten: no clearing
    if ((rc = px_composite_from_pert(ctx, img, mask_lg, pert, H, W, xf, att, st))) return rc;
    return px_warp_net(ctx, att, H, W, xf, 0, xf->T, 0, normalize, net, out, st);
}

extern "C" int gr_warp_batch(gr_ctx* ctx, const float* img, int H, int W, const float* masks, const float* thetas, int C, int nh,
                             int nw, const gr_xforms* xf, float lbd, int normalize, int net, float* out, void* stream) {
    NEED_CTX();
    if (C == 0) return GR_OK;                          // empty candidate list: nothing to do (pointers may be null)
    if (!img || !masks || !thetas || !out || C < 0) return gr_fail(ctx, GR_EINVAL, "gr_warp_batch: bad arguments");
    int rc = check_frame(ctx, H, W, nh, nw, net, xf);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const size_t per_cand = (size_t)xf->nslots * H * W * ATT_B;
    // candidates per pass bounded by ~2 GiB of composites
    int cpass = (int)(((size_t)2 << 30) / per_cand);
    if (cpass < 1) cpass = 1;
    if (cpass > C) cpass = C;
    if ((rc = gr_ws_reserve(ctx, gr_align(per_cand * cpass), stream))) return rc;
    float* att = (float*)ctx->ws;
    const int T = xf->T;
    for (int c0 = 0; c0 < C; c0 += cpass) {
        int nc = C - c0 < cpass ? C - c0 : cpass;
        if ((rc = px_composite(ctx, img, H, W, masks + (size_t)c0 * 3 * nh * nw, thetas + (size_t)c0 * 3 * nh * nw, nc, nh, nw,
                               lbd, xf, att, st))) return rc;
        long long nimg = (long long)nc * T;
        for (long long o = 0; o < nimg; o += (1 << 30)) {
            int cnt = (int)(nimg - o < (1 << 30) ? nimg - o : (1 << 30));
            if ((rc = px_warp_net(ctx, att, H, W, xf, (int64_t)c0 * T + o, cnt, c0, normalize, net,
                                  out + ((size_t)c0 * T + o) * 3 * net * net, st))) return rc;
        }
    }
    return GR_OK;
}

// ---------------------------------------------------------------------------------------------
// classifiers
// ---------------------------------------------------------------------------------------------
extern "C" int gr_model_forward(gr_ctx* ctx, gr_model* m, const float* x, int N, float* logits, void* stream) {
    NEED_CTX();
    if (!m || !x || !logits || N < 0) return gr_fail(ctx, GR_EINVAL, "gr_model_forward: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int ncls = gr_model_num_classes(m);
    int chunk = chunk_for(ctx, m, N);
    if (chunk <= 0) return GR_OK;
    size_t wsb = net_workspace_bytes(m, chunk);
    int rc = gr_ws_reserve(ctx, wsb, stream);
    if (rc) return rc;
    for (int n0 = 0; n0 < N; n0 += chunk) {
        int nb = N - n0 < chunk ? N - n0 : chunk;
        if ((rc = net_forward(ctx, m, x + (size_t)n0 * 3072, nb, logits + (size_t)n0 * ncls, ctx->ws, ctx->ws_bytes, st))) return rc;
    }
    return GR_OK;
}

extern "C" int gr_model_predict(gr_ctx* ctx, gr_model* m, const float* x, int N, int64_t* preds, void* stream) {
    NEED_CTX();
    if (!m || !x || !preds || N < 0) return gr_fail(ctx, GR_EINVAL, "gr_model_predict: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int ncls = gr_model_num_classes(m);
    int chunk = chunk_for(ctx, m, N);
    if (chunk <= 0) return GR_OK;
    size_t lg_bytes = gr_align((size_t)chunk * ncls * sizeof(float));
    size_t wsb = net_workspace_bytes(m, chunk);
    int rc = gr_ws_reserve(ctx, lg_bytes + wsb, stream);
    if (rc) return rc;
    float* lg = (float*)ctx->ws;
    for (int n0 = 0; n0 < N; n0 += chunk) {
        int nb = N - n0 < chunk ? N - n0 : chunk;
        if ((rc = net_forward(ctx, m, x + (size_t)n0 * 3072, nb, lg, ctx->ws + lg_bytes, ctx->ws_bytes - lg_bytes, st))) return rc;
        if ((rc = net_argmax(ctx, lg, nb, ncls, preds + n0, st, net_tie_last(m)))) return rc;
    }
    return GR_OK;
}

// ---- PatchGuard robust classifier (patchguard/patchguard_predict.py) ----
extern "C" int gr_pg_preprocess(gr_ctx* ctx, const float* x, int N, int S, int D, float* out, void* stream) {
    NEED_CTX();
    if (N == 0) return GR_OK;                          // empty batch: nothing to do (pointers may be null)
    if (!x || !out || N < 0) return gr_fail(ctx, GR_EINVAL, "gr_pg_preprocess: bad arguments");
    return pg_preprocess(ctx, x, N, S, D, false, nullptr, out, (hipStream_t)stream);
}
extern "C" int gr_pg_masking_defense(gr_ctx* ctx, const float* local, int N, int h, int w, int ncls, int window, float thres,
                                     float* scores, int64_t* preds, void* stream) {
    NEED_CTX();
    if (N == 0) return GR_OK;
    if (!local || N < 0 || (!scores && !preds)) return gr_fail(ctx, GR_EINVAL, "gr_pg_masking_defense: bad arguments");
    return pg_masking(ctx, local, N, h, w, ncls, window, window, thres, scores, preds, (hipStream_t)stream);
}
extern "C" int gr_pg_local_logits(gr_ctx* ctx, gr_model* m, const float* x, int N, float* local, void* stream) {
    NEED_CTX();
    if (m && N == 0) return GR_OK;
    if (!m || !x || !local || N < 0) return gr_fail(ctx, GR_EINVAL, "gr_pg_local_logits: bad arguments");
    const int h = net_pg_out_size(m);
    if (h <= 0) return gr_fail(ctx, GR_EINVAL, "gr_pg_local_logits: not a PatchGuard model");
    hipStream_t st = (hipStream_t)stream;
    const int ncls = gr_model_num_classes(m);
    int chunk = chunk_for(ctx, m, N);
    if (chunk <= 0) return GR_OK;
    int rc = gr_ws_reserve(ctx, net_workspace_bytes(m, chunk), stream);
    if (rc) return rc;
    for (int n0 = 0; n0 < N; n0 += chunk) {
        int nb = N - n0 < chunk ? N - n0 : chunk;
        if ((rc = net_pg_local(ctx, m, x + (size_t)n0 * 3072, nb, local + (size_t)n0 * h * h * ncls, ctx->ws, ctx->ws_bytes, st))) return rc;
    }
    return GR_OK;
}

extern "C" int gr_model_input_grad(gr_ctx* ctx, gr_model* m, const float* x, int N, int target, float* logits,
                                   float* grad_x, void* stream) {
    NEED_CTX();
    if (!m || !x || !grad_x || N < 0) return gr_fail(ctx, GR_EINVAL, "gr_model_input_grad: bad arguments");
    if (net_pg_out_size(m) > 0) return gr_fail(ctx, GR_EINVAL, "gr_model_input_grad: the PatchGuard classifier has no gradient path (the reference has none either)");
    hipStream_t st = (hipStream_t)stream;
    const int ncls = gr_model_num_classes(m);
    // the pass keeps every activation: ~0.8 MB per image for GTSRBNet, ~17 MB for WideResNet-34-10
    int64_t cap = (int64_t)(ctx->ws_limit / (net_grad_workspace_bytes(m, 64) / 64 + 1));
    if (cap < 1) cap = 1;
    if (cap > 1024) cap = 1024;
    int chunk = N < cap ? N : (int)cap;
    if (chunk <= 0) return GR_OK;
    int rc = gr_ws_reserve(ctx, net_grad_workspace_bytes(m, chunk), stream);
    if (rc) return rc;
    for (int n0 = 0; n0 < N; n0 += chunk) {
        int nb = N - n0 < chunk ? N - n0 : chunk;
        if ((rc = net_input_grad(ctx, m, x + (size_t)n0 * 3072, nb, target, logits ? logits + (size_t)n0 * ncls : nullptr,
                                 grad_x + (size_t)n0 * 3072, ctx->ws, ctx->ws_bytes, st))) return rc;
    }
    return GR_OK;
}

// ---------------------------------------------------------------------------------------------
// fused TR evaluation
// ---------------------------------------------------------------------------------------------
extern "C" int gr_tr_hits(gr_ctx* ctx, gr_model* m, const float* img, int H, int W, const float* masks, const float* thetas,
                          int C, int nh, int nw, const gr_xforms* xf, float lbd, int normalize, int target, int64_t img_begin,
                          int64_t img_count, int32_t* hits, int64_t* preds, float* logits, void* stream) {
    NEED_CTX();
    if (!m || !img || !masks || !thetas || !hits || C < 0) return gr_fail(ctx, GR_EINVAL, "gr_tr_hits: bad arguments");
    int rc = check_frame(ctx, H, W, nh, nw, 32, xf);
    if (rc) return rc;
    const int T = xf->T;
    const int64_t total = (int64_t)C * T;
    if (img_begin < 0 || img_count < 0 || img_begin + img_count > total) return gr_fail(ctx, GR_EINVAL, "gr_tr_hits: image slice out of range");
    hipStream_t st = (hipStream_t)stream;
    if (C > 0) GR_HIP(ctx, hipMemsetAsync(hits, 0, sizeof(int32_t) * C, st));
    if (img_count == 0) return GR_OK;
    const int ncls = gr_model_num_classes(m);
    const int chunk = chunk_for(ctx, m, img_count);
    // workspace: composites of the candidates one chunk can touch | chunk logits | net scratch.  The warp kernel writes
    // the classifier's NHWC4 input (GTSRBNet: after conv1_1 + ReLU) straight into the net scratch: no CHW image buffer,
    // no layout-change kernel in between.
    const int max_cand = (int)((chunk + T - 1) / T + 1);
    const size_t per_cand = (size_t)xf->nslots * H * W * ATT_B;
    const size_t att_bytes = gr_align(per_cand * (size_t)(max_cand < C ? max_cand : C));
    const size_t lg_bytes = gr_align((size_t)chunk * ncls * sizeof(float));
    const size_t net_bytes = net_workspace_bytes(m, chunk);
    if ((rc = gr_ws_reserve(ctx, att_bytes + lg_bytes + net_bytes, stream))) return rc;
    float* att = (float*)ctx->ws;
    float* lg = (float*)(ctx->ws + att_bytes);
    char* nws = ctx->ws + att_bytes + lg_bytes;
    const float* c11 = nullptr;
    const int in_mode = net_input_mode(m, &c11);
    int have_lo = -1, have_hi = -1;     // candidates whose composites are currently in `att`
    for (int64_t n0 = img_begin; n0 < img_begin + img_count; n0 += chunk) {
        int nb = (int)(img_begin + img_count - n0 < chunk ? img_begin + img_count - n0 : chunk);
        int c_lo = (int)(n0 / T), c_hi = (int)((n0 + nb - 1) / T);
        if (!(c_lo >= have_lo && c_hi <= have_hi)) {
            int nc = c_hi - c_lo + 1;
            if ((rc = px_composite(ctx, img, H, W, masks + (size_t)c_lo * 3 * nh * nw, thetas + (size_t)c_lo * 3 * nh * nw, nc,
                                   nh, nw, lbd, xf, att, st))) return rc;
            have_lo = c_lo; have_hi = c_hi;
        }
        if ((rc = px_warp_net(ctx, att, H, W, xf, n0, nb, have_lo, normalize, 32, net_input_x4(nws), st, in_mode, c11))) return rc;
        float* lg_out = logits ? logits + (size_t)n0 * ncls : lg;
        if ((rc = net_forward(ctx, m, nullptr, nb, lg_out, nws, net_bytes, st))) return rc;
        if ((rc = net_hits(ctx, lg_out, nb, ncls, n0, T, target, hits, preds ? preds + n0 : nullptr, st, net_tie_last(m)))) return rc;
    }
    return GR_OK;
}
