/*
 * graphite_hip.h -- C ABI of libgraphite_hip.so: GRAPHITE's transform-robustness (TR) scoring
 * path on MI355X (gfx950).
 *
 * The reference (ryan-feng/GRAPHITE, /root/reference) is pure Python and has no FFI of its own;
 * its de-facto boundary for this path is the pair
 *     transforms.get_transformed_images(...)   transforms.py:162
 *     utils.run_predictions(model, imgs, ...)  utils.py:60
 * plus the helpers they call.  Each entry point below names the reference function (file:line)
 * whose arithmetic it replaces.  The Python package graphite_amd/ binds these with ctypes and
 * re-exports the reference's own function names (INTEGRATION.md shows the binding).
 *
 * Conventions
 *  - Plain C: pointers + sizes only, no C++/torch types.  Returns 0 on success, a negative
 *    GR_E* code otherwise; gr_last_error() gives the message.  No exception crosses the ABI.
 *  - "dev" pointers are device (HBM) addresses on the context's GPU; "host" pointers are
 *    ordinary host memory.  The caller owns every buffer it passes; the library owns gr_ctx,
 *    gr_xforms and gr_model objects and a growable device workspace inside gr_ctx.
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Calls enqueue work
 *    on it and return without synchronising unless stated; one host thread per gr_ctx.
 *    Every call makes the context's device the calling thread's current HIP device.
 *  - Images are planar CHW float32.  Candidate c = (mask[c], theta[c]); image n = c*T + t.
 */
#ifndef GRAPHITE_HIP_H
#define GRAPHITE_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GR_OK 0
#define GR_EINVAL (-1)   /* bad argument (null pointer, bad size, unsupported blur, ...) */
#define GR_EHIP (-2)     /* a HIP runtime call failed */
#define GR_ENOMEM (-3)   /* device workspace allocation failed */
#define GR_ESTATE (-4)   /* object used with the wrong context / not initialised */

typedef struct gr_ctx gr_ctx;
typedef struct gr_model gr_model;
typedef struct gr_xforms gr_xforms;

/* One sampled transform, as the reference computes it per call:
 *   M, crop   perspective_transform.py:11-44 (forward src->dst matrix and dsize given to
 *             cv2.warpPerspective, :47)
 *   blur      transforms.py:55-56 (Gaussian kernel size 0/3/5/7)
 *   lut       transforms.py:70-72 (gamma table) */
typedef struct gr_xform {
    double M[9];
    int32_t crop_x, crop_y;
    int32_t blur;
    int32_t pad_;
    uint8_t lut[256];
} gr_xform;

/* ---- context ------------------------------------------------------------------------------- */
int gr_ctx_create(int device, gr_ctx** out);
void gr_ctx_destroy(gr_ctx* ctx);
/* Message of the last failing call on ctx (ctx == NULL: last gr_ctx_create failure). */
const char* gr_last_error(const gr_ctx* ctx);
/* Images classified per internal chunk of gr_tr_hits (default 8192; bounds the workspace). */
int gr_ctx_set_chunk(gr_ctx* ctx, int images_per_chunk);
/* Bytes of device workspace currently held. */
int64_t gr_ctx_workspace_bytes(const gr_ctx* ctx);
/* Upper bound for the classifier scratch of one chunk (default 8 GiB): the chunk of a call is
 * min(gr_ctx_set_chunk's value, limit / scratch bytes per image of the model) -- GTSRBNet keeps 176 KB per image,
 * WideResNet-34-10 3.3 MB, so its chunks are 2,500 images at the default.  The workspace is grow-only and is returned
 * by gr_ctx_destroy. */
int gr_ctx_set_workspace_limit(gr_ctx* ctx, int64_t bytes);

/* ---- measurement (no counterpart in the reference, which only prints time.time() deltas,
 * boost.py:27,55; coarse_reduction.py:76).  With profiling on, every kernel launch the library
 * makes is bracketed by a HIP event pair on the caller's stream; gr_profile_read synchronises on
 * the recorded events and returns, per kernel, launches, total milliseconds and the ALGORITHMIC
 * flops / bytes of those launches (DESIGN.md states the per-unit figures). */
typedef struct gr_prof_entry {
    char name[64];
    int64_t launches;
    double ms, flops, bytes;
} gr_prof_entry;
int gr_profile_enable(gr_ctx* ctx, int on);
int gr_profile_reset(gr_ctx* ctx);
/* Returns the number of entries written (out == NULL: the number available). */
int gr_profile_read(gr_ctx* ctx, int max_entries, gr_prof_entry* out);

/* ---- transform table ------------------------------------------------------------------------
 * Uploads T descriptors once (inverse matrices are formed in double on the host, as
 * cv2.warpPerspective does before warping); reused by every call of an attack. */
int gr_xforms_create(gr_ctx* ctx, const gr_xform* host_xf, int T, gr_xforms** out);
void gr_xforms_destroy(gr_xforms* xf);

/* ---- pixel operators (all pointers dev) ----------------------------------------------------- */
/* cv2.resize(src,(dw,dh)) INTER_LINEAR on C planes: transforms.py:23,148,151; boost.py:33,42. */
int gr_resize_linear(gr_ctx* ctx, const float* src, int C, int sh, int sw, float* dst, int dh, int dw, void* stream);
/* convert2Network.  net == 32 (transforms.py:18-29): resize to 32 x 32, clamp [0,1], minus 0.5 if normalize;
 * dst = [3,32,32].  net != 32 (transforms.py:31-45, the licence-plate branch): the short side becomes net, the long side
 * int(round(net / short * long)) (gr_net_size), clamp only -- `normalize` is ignored as that branch never normalises;
 * dst = [3,ny,nx]. */
int gr_convert_to_network(gr_ctx* ctx, const float* src, int sh, int sw, float* dst, int net, int normalize,
                          void* stream);
/* add_noise, transforms.py:143-159.  comb / pert / mask_lg are [3,H,W] outputs, any may be NULL. */
int gr_add_noise(gr_ctx* ctx, const float* img, int H, int W, const float* mask, const float* theta, int nh, int nw,
                 float lbd, int clip, float* comb, float* pert, float* mask_lg, void* stream);
/* cv2.warpPerspective(img, M, (crop_x,crop_y), BORDER_REPLICATE), perspective_transform.py:47.
 * M = 9 host doubles (forward matrix); dst = [3,crop_y,crop_x]. */
int gr_warp_perspective(gr_ctx* ctx, const float* src, int H, int W, const double* host_M, int crop_x, int crop_y,
                        float* dst, void* stream);
/* apply_transformation, transforms.py:50-77, for T transforms of one already-composited triple
 * (img, mask_lg, pert as returned by add_noise).  out = [T,3,net,net]. */
int gr_transform_composited(gr_ctx* ctx, const float* img, const float* mask_lg, const float* pert, int H, int W,
                            const gr_xforms* xf, int normalize, int net, float* out, void* stream);
/* Output size of convert2Network for a warped image of crop_x x crop_y (host helper, no GPU work): 32 x 32 for
 * net_size == 32 (transforms.py:19-23), else transforms.py:35-40 in Python's float arithmetic (round half to even). */
int gr_net_size(int crop_x, int crop_y, int net_size, int* nx, int* ny);
/* The net_size != 32 branch (transforms.py:31-45 behind apply_transformation :77 / get_transformed_images :162-173) for
 * frames of any aspect ratio (perspective_transform.py:39-44 gives non-square dsize): every transform t yields its OWN
 * [3, ny_t, nx_t] image (gr_net_size of its crop), clamped to [0,1], not normalised.  Image i (= t, or c*T + t) is
 * written at out + i * out_stride floats; out_stride >= the largest 3*ny*nx of the table (GR_EINVAL otherwise).
 * The reference returns these as a Python list of differently sized tensors. */
int gr_transform_composited_var(gr_ctx* ctx, const float* img, const float* mask_lg, const float* pert, int H, int W,
                                const gr_xforms* xf, int net_size, float* out, int64_t out_stride, void* stream);
int gr_warp_batch_var(gr_ctx* ctx, const float* img, int H, int W, const float* masks, const float* thetas, int C, int nh,
                      int nw, const gr_xforms* xf, float lbd, int net_size, float* out, int64_t out_stride, void* stream);
/* gr_transform_composited for callers that cannot allocate device memory (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arguments, same results bit for bit;
 * the composites (one RGB f32 image, 12-byte texels, per blur kernel in use, <= 4) are kept in a file-scope __device__ array of
 * 4 * GR_STATIC_MAX_PIXELS * 12 bytes (3 MiB) that every GPU the library runs on holds until the process ends, instead of
 * the growable workspace: the call makes no hipMalloc / hipFree and creates no events.  GR_EINVAL when H*W exceeds
 * GR_STATIC_MAX_PIXELS (244x244 GTSRB frames fit); GR_ESTATE while gr_profile_enable is on.  All calls on one GPU share
 * that array: calls that overlap on one GPU (two streams, two contexts on the same device) race.  Nothing in it has to
 * start at zero -- every texel a call reads it has written first. */
#define GR_STATIC_MAX_PIXELS 65536
int gr_transform_composited_static(gr_ctx* ctx, const float* img, const float* mask_lg, const float* pert, int H, int W,
                                   const gr_xforms* xf, int normalize, int net, float* out, void* stream);
/* get_transformed_images, transforms.py:162-173, for C candidates x T transforms in one pass:
 * masks/thetas = [C,3,nh,nw]; out = [C,T,3,net,net]. */
int gr_warp_batch(gr_ctx* ctx, const float* img, int H, int W, const float* masks, const float* thetas, int C,
                  int nh, int nw, const gr_xforms* xf, float lbd, int normalize, int net, float* out, void* stream);

/* ---- victim classifiers ---------------------------------------------------------------------
 * GTSRBNet, GTSRB/GTSRBNet.py:8-52.  tensors = 20 host pointers: weight, bias of conv1_1,
 * conv2_1, conv2_2, conv3_1, conv3_2, conv4_1, conv4_2, fc5_1, fc6_1, fc7_1 in PyTorch layouts
 * (OIHW / [out,in]); numel[i] is checked against the architecture. */
int gr_model_load_gtsrb(gr_ctx* ctx, const float* const* tensors, const int64_t* numel, int n, gr_model** out);
/* WideResNet, cifar/wideresnet.py:52-94.  tensors (host): conv1.weight; for block1..3, layer
 * 0..n-1: bn1.{weight,bias,running_mean,running_var}, conv1.weight, bn2.{same four},
 * conv2.weight, [convShortcut.weight when in != out planes]; then bn1.{four}, fc.weight, fc.bias. */
int gr_model_load_wrn(gr_ctx* ctx, const float* const* tensors, const int64_t* numel, int n, int depth, int widen,
                      int num_classes, gr_model** out);
void gr_model_destroy(gr_model* m);
int gr_model_num_classes(const gr_model* m);
/* Arithmetic of the conv layers.  GR_PREC_F32 (default): fp32 MFMA, the parity path (labels, counts and logits
 * bit-exact against the oracle).  GR_PREC_BF16: the north star's throughput mode -- GTSRBNet's conv2_1 .. conv4_2
 * and fc5 / fc6 (all but 0.03 % of the flops) take bf16 operands (activations and weights rounded to nearest even) with f32
 * accumulation; conv1_1 and fc7 stay fp32.  Not bit-exact.  Each layer alone follows its restatement (torch,
 * .bfloat16() operands, wide accumulation) to a median 1e-5 of the largest logit; end to end, activations that land on a
 * bf16 rounding boundary round differently in any two implementations, so logits agree to ~1e-2 of the largest logit --
 * the same order as bf16 against fp32 itself (tests/test_gpu_bf16.py has the numbers).  WideResNet: the 3x3 convs (stride 1
 * and 2) with Cin % 32 == 0, Cout % 160 == 0 (95 % of WRN-34-10's flops) run in bf16, the rest stays fp32; GR_EINVAL for a net without
 * such a layer.  PatchGuard (gr_model_load_bagnet): BagNet's stride-1 1x1 convs (74 % of bagnet17's flops) run as bf16 GEMMs,
 * activations stay f32 in HBM; 3x3 convs, strided shortcuts, the fc and the masking epilogue stay fp32. */
#define GR_PREC_F32 0
#define GR_PREC_BF16 1
int gr_model_set_precision(gr_model* m, int precision);
int gr_model_get_precision(const gr_model* m);
/* Which GTSRBNet layers GR_PREC_BF16 moves to bf16: bit 0 = conv2_1, 1 = conv2_2, 2 = conv3_1, 3 = conv3_2, 4 = conv4_1,
 * 5 = conv4_2, 6 = fc5 + fc6 (default 0x7f = all of them; the activations conv4_1 -> conv4_2 -> fc5 -> fc6 are then stored
 * as bf16, which does not change a single logit: bit 7 keeps them f32 so that a test can show it).  For tests and error budgets: one layer at a time can be compared with its
 * restatement without the rounding-boundary flips of the layers before it. */
int gr_model_set_bf16_layers(gr_model* m, unsigned mask);
/* forward: x = [N,3,32,32] dev (already in the net's input range) -> logits [N,ncls] dev. */
int gr_model_forward(gr_ctx* ctx, gr_model* m, const float* x, int N, float* logits, void* stream);
/* predict(): torch.max(logits,1) first-maximum rule (GTSRBNet.py:66) -> preds [N] int64 dev. */
int gr_model_predict(gr_ctx* ctx, gr_model* m, const float* x, int N, int64_t* preds, void* stream);

/* ---- PatchGuard robust classifier, patchguard/patchguard_predict.py:21-89 (BASELINE config 5) --------------------------
 * forward (:38-50):  z = (x + 0.5) * 255;  z = F.interpolate(z, 192, mode='bicubic') / 255;  z = Normalize(mean, std)(z)
 *                    local = bagnet17(aggregation='none')(z)                       [B, 22, 22, 10] local logits
 *                    pred_i = masking_defense(local[i], thres=0.0, window_shape=[6, 6])
 * Lines 42-44 and the Normalize of :30-32 are REFERENCE code (torch's upsample_bicubic2d, A = -0.75, half-pixel centres,
 * clamped border taps); BagNet-17 (PatchGuard.nets.bagnet) and masking_defense (PatchGuard.utils.defense_utils) are
 * THIRD-PARTY code of inspire-group/PatchGuard @ ae23629 (README.md:113-117) whose source and checkpoint are absent:
 * both are restated from the publications (BagNet: Brendel & Bethge, ICLR 2019; masking: PatchGuard Algorithm 1) --
 * "parity unpinned" for those two.
 *
 * gr_pg_preprocess (:42-44, :30-32): x [N,3,S,S] dev in [-0.5,0.5] -> out [N,3,D,D] dev, normalised. */
int gr_pg_preprocess(gr_ctx* ctx, const float* x, int N, int S, int D, float* out, void* stream);
/* BagNet (:24-29).  tensors (host, PyTorch layouts): conv1.weight [64,3,1,1], conv2.weight [64,64,3,3], bn1.{weight, bias,
 * running_mean, running_var}; per stage g = 0..3 (planes 64 << g) and block i < layers[g]: conv1.weight, bn1.{4},
 * conv2.weight (k x k, k = 3 iff kernel3[g] > i, stride strides[g] in block 0), bn2.{4}, conv3.weight, bn3.{4},
 * [downsample.0.weight, downsample.1.{4} in block 0]; fc.weight [ncls, 2048], fc.bias.  bagnet17: layers {3,4,6,3},
 * strides {2,2,2,1}, kernel3 {1,1,1,0}.  `upsample` = the bicubic target size (192, :43).  The model's input is
 * [N,3,32,32] in [-0.5,0.5] as for the other classifiers; gr_model_forward returns [N,ncls] = the masked class evidence
 * (defense on, the default) or the mean local logits (PatchGuardNoDefense, :56-89), gr_model_predict / gr_tr_hits take
 * the LAST maximum with the defense on (np.argsort(kind='stable')[-1]) and the first otherwise (np.argmax). */
int gr_model_load_bagnet(gr_ctx* ctx, const float* const* tensors, const int64_t* numel, int n, const int* layers,
                         const int* strides, const int* kernel3, int num_classes, int upsample, gr_model** out);
/* defense != 0: masking_defense with this window (default 6 = ceil((30 + 17 - 1) / 8), :33-36) and threshold (0.0, :48);
 * defense == 0: mean aggregation. */
int gr_pg_configure(gr_model* m, int defense, int window, float thres);
/* side h of the local-logit map (22 for bagnet17 at 192); 0 for other models */
int gr_pg_out_size(const gr_model* m);
/* local logits (:44): x [N,3,32,32] dev -> local [N,h,h,ncls] dev */
int gr_pg_local_logits(gr_ctx* ctx, gr_model* m, const float* x, int N, float* local, void* stream);
/* masking_defense (:48) on caller-supplied local logits [N,h,w,ncls] dev (any local-logit net can sit in front):
 * scores [N,ncls] dev = masked class evidence, preds [N] int64 dev; either may be NULL.  Maps of up to 64 KB per image
 * are staged in LDS, larger ones are read from global memory (same sums, same order). */
int gr_pg_masking_defense(gr_ctx* ctx, const float* local, int N, int h, int w, int ncls, int window, float thres,
                          float* scores, int64_t* preds, void* stream);

/* Input gradient of the classifier for the whitebox EOT loop (whitebox_attack.py:155-158:
 * logits = model(x); loss = CrossEntropyLoss(logits, target); grad = autograd.grad(loss, x)):
 * grad_x [N,3,32,32] dev = d CE(logits_n, target) / d x_n PER IMAGE (no averaging over N); logits [N,ncls] dev optional.
 * GTSRBNet: forward with every activation kept (GTSRBNet.py:35-52), then the transposed layers on the same MFMA kernels
 * (flipped 5x5 taps, W^T), ReLU / max-pool routing as torch's autograd does (first maximum of a window).
 * WideResNet (cifar/wideresnet.py:25-34, 86-94): every post-activation kept, then per block in reverse the transposed 3x3
 * convs (strided layers on the zero-dilated gradient), eval-BatchNorm scale and ReLU gates, 1x1 / identity shortcuts,
 * average-pool and linear backward.  ~17 MB of workspace per image for WRN-34-10 (chunked under the workspace limit).
 * Always fp32, whatever gr_model_set_precision says. */
int gr_model_input_grad(gr_ctx* ctx, gr_model* m, const float* x, int N, int target, float* logits, float* grad_x,
                        void* stream);

/* ---- fused TR evaluation --------------------------------------------------------------------
 * run_predictions(get_transformed_images(...)) for C candidates at once (utils.py:29-57):
 * hits[c] = #{t : argmax(model(image(c,t))) == target}.  hits = int32[C] dev (overwritten).
 * preds (int64[C*T]) and logits (float[C*T,ncls]) are optional dev outputs (NULL to skip).
 * img_begin/img_count select the contiguous slice of the flattened image index n = c*T + t that
 * this call (this rank) evaluates (0, C*T = everything); hits[] holds the partial counts of that
 * slice and 0 elsewhere, so the ranks' vectors can be summed (one int32[C] all-reduce).
 * preds/logits, when given, are full-size [C*T] arrays of which only the slice is written. */
int gr_tr_hits(gr_ctx* ctx, gr_model* m, const float* img, int H, int W, const float* masks, const float* thetas,
               int C, int nh, int nw, const gr_xforms* xf, float lbd, int normalize, int target, int64_t img_begin,
               int64_t img_count, int32_t* hits, int64_t* preds, float* logits, void* stream);

/* ---- differentiable ("whitebox") transform, forward and backward ------------
 * transform_wb (transforms.py:176-207) for T transforms of ONE (orig, att, mask) at once:
 *   clamp(att) -> [depthwise Gaussian conv, zero padded, of the masked perturbation] -> where(mask) -> clamp ->
 *   kornia warp_perspective(align_corners=True, padding_mode='border') (perspective_transform.py:49) ->
 *   adjust_gamma -> resize(net, align_corners=False) -> clamp                       (all float32)
 * One descriptor per transform, prepared on the host as kornia's normalize_homography does:
 *   A      float32 3x3 mapping NORMALISED dst coordinates to NORMALISED src coordinates
 *          (inverse of N_dst * M * inv(N_src), in float32)
 *   crop   (crop_x, crop_y) = dsize of the warp (both > 1); gamma; blur in {0,3,5,7}.
 * orig, att, mask: [3,H,W] float32 dev (mask 0/1).  xf: HOST array of T descriptors (copied per call;
 * the call synchronises the stream once for that copy).
 * gr_wb_forward : out      [T,3,net,net] dev, values in [0,1] (the caller subtracts 0.5 for GTSRB, whitebox_attack.py:155).
 * gr_wb_backward: grad_out [T,3,net,net] dev = d loss / d out for all T; grad_att [3,H,W] dev (overwritten) =
 *                 d loss / d att summed over the T transforms (the EOT loop averages it, whitebox_attack.py:159-162).
 *                 Accumulation uses float atomics: the summation order, hence the last bits, vary between runs.
 *                 pow's derivative at att == 0 follows IEEE (inf / nan for gamma < 1), as torch's does. */
typedef struct gr_wb_xform {
    float A[9];
    int32_t crop_x, crop_y;
    float gamma;
    int32_t blur;
} gr_wb_xform;
int gr_wb_forward(gr_ctx* ctx, const float* orig, const float* att, const float* mask, int H, int W,
                  const gr_wb_xform* xf, int T, int net, float* out, void* stream);
int gr_wb_backward(gr_ctx* ctx, const float* orig, const float* att, const float* mask, int H, int W,
                   const gr_wb_xform* xf, int T, int net, const float* grad_out, float* grad_att, void* stream);

#ifdef __cplusplus
}
#endif
#endif
