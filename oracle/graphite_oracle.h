/*
 * graphite_oracle.h -- CPU restatement of GRAPHITE's transform-robustness scoring path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under graphite_amd/ (the product) may include,
 * link, load or call this; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, as the checker / the timed CPU baseline.
 *
 * PARITY STATUS
 *  - Host geometry (transform sampling, homography, crop), score_fn, run_predictions
 *    batching and the classifier layers are PINNED against outputs of the reference's
 *    own Python run in the build container (tests/golden/make_golden.py).
 *  - The pixel arithmetic the reference delegates to opencv-python==4.2.0.32
 *    (requirements.txt:4; cv2.resize, cv2.GaussianBlur, cv2.warpPerspective, cv2.LUT --
 *    transforms.py:23,56,73,148,151; perspective_transform.py:47) is restated from the
 *    published OpenCV 4.2 algorithm.  OpenCV is neither vendored nor installed and the
 *    reference has no tests or pixel fixtures: PARITY UNPINNED at that boundary.
 *
 * All images are planar CHW float32 (the reference converts CHW<->HWC around every
 * cv2 call; per-channel arithmetic is identical).
 */
#ifndef GRAPHITE_ORACLE_H
#define GRAPHITE_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same layout as gr_xform in include/graphite_hip.h. */
typedef struct or_xform {
    double M[9];            /* FORWARD src->dst homography, as passed to cv2.warpPerspective */
    int32_t crop_x, crop_y; /* dsize */
    int32_t blur;           /* 0,3,5,7 */
    int32_t pad_;
    uint8_t lut[256];       /* gamma table, transforms.py:70-72 */
} or_xform;

/* cv2.resize(src,(dw,dh)) INTER_LINEAR on C planes. */
int or_resize_linear(const float* src, int C, int sh, int sw, float* dst, int dh, int dw);

/* transforms.py:143-159.  Any of comb/pert/mask_lg may be NULL. */
int or_add_noise(const float* img, int H, int W, const float* mask, const float* theta, int nh, int nw,
                 float lbd, int clip, float* comb, float* pert, float* mask_lg);

/* transforms.py:55-62: GaussianBlur(pert,(k,k),0) -> where(mask_lg>0.5, pert, img) -> clip. */
int or_composite(const float* img, const float* pert, const float* mask_lg, int H, int W, int blur, float* att);

/* perspective_transform.py:47 + transforms.py:69-77: warpPerspective(BORDER_REPLICATE) -> u8 -> LUT
 * -> /255 -> resize(net,net) -> clamp -> (-0.5 if normalize).  out = [3,net,net].
 * If warped != NULL it receives the full [3,crop_y,crop_x] warp (before the uint8 stage). */
int or_warp_one(const float* att, int H, int W, const or_xform* xf, int normalize, int net, float* out,
                float* warped);

/* get_transformed_images for C candidates x T transforms: out = [C,T,3,net,net]. */
int or_warp_batch(const float* img, int H, int W, const float* masks, const float* thetas, int C, int nh, int nw,
                  const or_xform* xf, int T, float lbd, int normalize, int net, float* out, int nthreads);

/* convert2Network's output size (transforms.py:19-23 for net_size 32, :35-40 otherwise). */
void or_net_size(int crop_x, int crop_y, int net_size, int* nx, int* ny);
/* The net_size != 32 branch (transforms.py:31-45) for frames of any aspect ratio (perspective_transform.py:39-44):
 * out = [3, ny, nx] per transform, clamped, never normalised; or_warp_batch_var places image (c, t) at
 * out + (c*T + t) * out_stride. */
int or_warp_one_var(const float* att, int H, int W, const or_xform* xf, int net_size, float* out);
int or_warp_batch_var(const float* img, int H, int W, const float* masks, const float* thetas, int C, int nh, int nw,
                      const or_xform* xf, int T, float lbd, int net_size, float* out, long long out_stride, int nthreads);

/* GTSRBNet (GTSRB/GTSRBNet.py:35-52).  tensors = 20 pointers in state_dict order
 * (conv1_1.weight, conv1_1.bias, ... fc7_1.weight, fc7_1.bias), PyTorch layouts. */
int or_gtsrb_forward(const float* const* tensors, const float* x, int N, float* logits, int nthreads);

/* WideResNet (cifar/wideresnet.py:86-94).  tensors in the order documented in
 * include/graphite_hip.h (gr_model_load_wrn). */
int or_wrn_forward(const float* const* tensors, int ntensors, int depth, int widen, int num_classes,
                   const float* x, int N, float* logits, int nthreads);

/* The same forward, also writing every ReLU decision (1 = active; or_wrn_mask_bytes(depth, widen) bytes per image: per block
 * relu(bn1(x)) [H,H,cin] then relu(bn2(conv1)) [Ho,Ho,cout], last the final relu(bn1) [8,8,ch3], all NHWC).  The HIP
 * library's forward is this forward bit for bit, so these are its decisions (tests/test_gpu_wb.py: gradient evidence). */
long long or_wrn_mask_bytes(int depth, int widen);
int or_wrn_forward_masks(const float* const* tensors, int ntensors, int depth, int widen, int num_classes,
                         const float* x, int N, float* logits, int nthreads, uint8_t* masks);

/* torch.max(out,1) first-max argmax; hits[c] += (pred == target) over the T images of candidate c. */
int or_argmax_hits(const float* logits, int N, int ncls, int T, int target, int64_t* preds, int32_t* hits);

/* PatchGuard robust classifier, patchguard/patchguard_predict.py:21-53 (see graphite_oracle.c for what is reference code
 * and what is restated third-party code).
 * or_pg_preprocess: x [N,3,S,S] in [-0.5,0.5] -> (x+0.5)*255 -> torch bicubic to D x D -> /255 -> Normalize -> [N,3,D,D].
 * or_bagnet_forward: tensors (PyTorch layouts) = conv1.weight, conv2.weight, bn1.{weight,bias,running_mean,running_var};
 *   per stage g = 0..3, block i: conv1.weight, bn1.{4}, conv2.weight, bn2.{4}, conv3.weight, bn3.{4},
 *   [downsample.0.weight, downsample.1.{4} for i == 0]; fc.weight, fc.bias.  z [N,3,D,D] -> local [N,h,h,ncls],
 *   h = or_bagnet_out_size(D, ...).
 * or_masking_defense / or_pg_mean_logits: local [N,h,w,ncls] -> scores [N,ncls] (+ preds with the last-maximum rule of
 *   np.argsort(kind='stable')[-1]; the no-defense variant is argmax'd with the first-maximum rule by or_argmax_hits). */
int or_pg_preprocess(const float* x, int N, int S, int D, float* out, int nthreads);
int or_bagnet_out_size(int D, const int* layers, const int* strides, const int* kernel3);
int or_bagnet_forward(const float* const* tensors, int ntensors, const int* layers, const int* strides, const int* kernel3,
                      int num_classes, const float* z, int N, int D, float* local, int nthreads);
int or_masking_defense(const float* local, int N, int h, int w, int ncls, int wx, int wy, float thres, float* scores,
                       int64_t* preds);
int or_pg_mean_logits(const float* local, int N, int h, int w, int ncls, float* scores);

/* Margin probes, never used by the parity tests proper (tests/pixel_margin.py): move every warped value n ulps before the
 * uint8 truncation of transforms.py:69; and whether this library is the FMA-contracted build (libgraphite_oracle_fma.so:
 * the same source compiled with -ffp-contract=fast, as OpenCV's AVX2 v_muladd / v_fma filter and resize paths round). */
void or_set_warp_nudge(int n);
int or_contracted_build(void);

#ifdef __cplusplus
}
#endif
#endif
