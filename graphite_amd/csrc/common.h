// Internal definitions shared by the translation units of libgraphite_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <array>
#include <atomic>
#include <string>
#include <vector>
#include "graphite_hip.h"

// Optional per-kernel timing with HIP events on the caller's stream (gr_profile_*).
struct ProfEntry { std::string name; long long launches = 0; double ms = 0, flops = 0, bytes = 0; };
struct ProfPending { int entry; hipEvent_t e0, e1; };

struct gr_ctx {
    int device = 0;
    std::string err;
    char* ws = nullptr;          // device workspace, grown on demand
    size_t ws_bytes = 0;
    int chunk = 8192;            // images per classifier chunk in gr_tr_hits
    size_t ws_limit = (size_t)8 << 30;   // the classifier scratch of one chunk stays under this (gr_ctx_set_workspace_limit)
    char* static_scratch = nullptr;   // this device's copy of api.hip's file-scope `scratch` array (gr_ctx_create)
    bool prof = false;
    std::vector<ProfEntry> prof_entries;
    std::vector<ProfPending> prof_pending;
    std::vector<hipEvent_t> prof_pool;
};

// RAII: records an event pair around the launches made in its scope when profiling is on.
struct ProfScope {
    gr_ctx* ctx; hipStream_t st; hipEvent_t e0 = nullptr, e1 = nullptr; int entry = -1;
    ProfScope(gr_ctx* c, const char* name, hipStream_t s, double flops, double bytes);
    ~ProfScope();
};

// Device-side transform descriptor: inverse matrix (double), crop, blur slot, block width of the
// cv::warpPerspective tiling (coordinates are computed relative to the block start), gamma LUT.
struct XformDev {
    double Mi[9];
    int32_t crop_x, crop_y;
    int32_t slot;                // index of the composite (blur variant) this transform samples
    int32_t bw0;
    int32_t sep;                 // 1: Mi[1] == 0 and Mi[7] == 0 (every rotation about the y axis, perspective_transform.py:16-29):
                                 //    source x and the projective divide depend on the destination COLUMN only
    int32_t pad_;
    uint8_t lut[256];
};

// one workgroup of k_warp_band: transforms order[first .. first + n) of one composite slot
struct WarpGroup { int32_t slot, first, n, pad_; };
constexpr int WARP_BAND_K = 16;

struct gr_xforms {
    gr_ctx* ctx = nullptr;
    int T = 0;
    XformDev* dev = nullptr;
    int nslots = 0;              // number of distinct blur kernels in the table
    int slot_blur[4] = {0, 0, 0, 0};   // blur kernel size of each slot
    std::vector<std::array<int, 2>> crop;   // host copy of (crop_x, crop_y) per transform
    // k_warp_band: transforms sorted by composite slot and cut into groups of <= WARP_BAND_K (one workgroup each)
    int* order_dev = nullptr;    // [T] transform indices, slot-major
    WarpGroup* groups_dev = nullptr;
    int ngroups = 0;
    bool all_sep = false;        // every transform has Mi[1] == Mi[7] == 0
};

int gr_fail(gr_ctx* ctx, int code, const char* fmt, ...);
int gr_ws_reserve(gr_ctx* ctx, size_t bytes, void* stream);

#define GR_HIP(ctx, call)                                                                          \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) return gr_fail(ctx, GR_EHIP, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)

static inline size_t gr_align(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// Composite texels: the three channels of a pixel packed as 12 bytes (a texel is what one gather of the warp kernels
// fetches; without a pad float a sampled source row drags a quarter fewer cache lines through L1).
constexpr int ATT_C = 3;                     // floats per composite texel
constexpr int ATT_B = ATT_C * 4;             // bytes per composite texel

// ---- pixel.hip (launchers used by api.hip) ----
int px_resize(gr_ctx*, const float* src, int C, int sh, int sw, float* dst, int dh, int dw, int clamp, int normalize,
              hipStream_t);
int px_add_noise(gr_ctx*, const float* img, int H, int W, const float* mask, const float* theta, int nh, int nw,
                 float lbd, int clip, float* comb, float* pert, float* mask_lg, hipStream_t);
int px_warp_full(gr_ctx*, const float* src, int H, int W, const double* Mi, int crop_x, int crop_y, float* dst,
                 hipStream_t);
// composites for `ncand` candidates: att = [ncand][nslots][H][W][3] f32 (RGB, ATT_B bytes per texel)
int px_composite(gr_ctx*, const float* img, int H, int W, const float* masks, const float* thetas, int ncand, int nh,
                 int nw, float lbd, const gr_xforms* xf, float* att, hipStream_t);
int px_composite_from_pert(gr_ctx*, const float* img, const float* mask_lg, const float* pert, int H, int W,
                           const gr_xforms* xf, float* att, hipStream_t);
// images n in [n_begin, n_begin+count): candidate n/T (composite index n/T - cand0), transform n%T.
// out_mode GR_OUT_CHW: out = [count,3,net,net] CHW f32 (the reference's network input).
// GR_OUT_NHWC4 (net == 32 only): out = [count,32,32,4] (r,g,b,0), the classifiers' input layout;
// GR_OUT_NHWC4_C11 (net == 32 only): the same after GTSRBNet's conv1_1 (1x1, c11 = 9 weights + 3 biases, dev) + ReLU.
enum { GR_OUT_CHW = 0, GR_OUT_NHWC4 = 1, GR_OUT_NHWC4_C11 = 2 };
int px_warp_net(gr_ctx*, const float* att, int H, int W, const gr_xforms* xf, int64_t n_begin, int count, int cand0,
                int normalize, int net, float* out, hipStream_t, int out_mode = GR_OUT_CHW, const float* c11 = nullptr,
                int64_t var_stride = 0);
// var_stride > 0: the net_size != 32 branch of convert2Network (transforms.py:31-45): image i is written at
// out + i * var_stride as [3, ny, nx] with (nx, ny) = px_net_size(crop_x, crop_y, net); clamp only, no normalisation
void px_net_size(int crop_x, int crop_y, int net_size, int* nx, int* ny);

// ---- pg.hip (PatchGuard robust classifier: pre-processing and aggregation kernels) ----
// x: [N,3,S,S] CHW, or [N,S,S,4] when in_nhwc4.  conv1_w == nullptr: out = [N,3,D,D] CHW normalised;
// else out = [N,D,D,64] = BagNet conv1 (1x1, 3 -> 64, weights [64][3] dev) of the normalised image.
int pg_preprocess(gr_ctx*, const float* x, int N, int S, int D, bool in_nhwc4, const float* conv1_w, float* out, hipStream_t);
int pg_masking(gr_ctx*, const float* local, int N, int h, int w, int ncls, int wx, int wy, float thres, float* scores,
               int64_t* preds, hipStream_t);
int pg_mean(gr_ctx*, const float* local, int N, int h, int w, int ncls, float* scores, hipStream_t);

// ---- nets.hip ----
struct gr_model;
// x_chw == nullptr: the NHWC4 input (net_input_mode's layout) is already in place at net_input_x4(ws)
int net_forward(gr_ctx*, gr_model*, const float* x_chw, int N, float* logits, char* ws, size_t ws_bytes, hipStream_t);
// where and how a fused producer (the warp kernel) hands the classifier its input: [N,32,32,4] floats at the start of ws;
// mode = GR_OUT_NHWC4_C11 with *c11 set for GTSRBNet (conv1_1 + ReLU applied by the producer), GR_OUT_NHWC4 for WideResNet
static inline float* net_input_x4(char* ws) { return (float*)ws; }
int net_input_mode(const gr_model*, const float** c11);
size_t net_workspace_bytes(const gr_model*, int N);
int net_argmax(gr_ctx*, const float* logits, int N, int ncls, int64_t* preds, hipStream_t, int tie_last = 0);
// tie_last: the LAST maximum wins (np.argsort(kind='stable')[-1], PatchGuard's masking_defense) instead of torch.max's first
int net_hits(gr_ctx*, const float* logits, int N, int ncls, int64_t n_begin, int T, int target, int32_t* hits,
             int64_t* preds, hipStream_t, int tie_last = 0);
int net_tie_last(const gr_model*);
// PatchGuard models: local logits [N,h,h,ncls] of N [3,32,32] inputs (x_chw == nullptr: NHWC4 input in place)
int net_pg_local(gr_ctx*, gr_model*, const float* x_chw, int N, float* local, char* ws, size_t ws_bytes, hipStream_t);
int net_pg_out_size(const gr_model*);
// d CE(logits_n, target) / d x_n (GTSRBNet): forward with every activation kept, then the transposed layers
int net_input_grad(gr_ctx*, gr_model*, const float* x_chw, int N, int target, float* logits_or_null, float* grad_x, char* ws,
                   size_t ws_bytes, hipStream_t);
size_t net_grad_workspace_bytes(const gr_model*, int N);
