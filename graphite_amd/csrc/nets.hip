// nets.hip -- victim classifiers of the transform-robustness path on gfx950:
//   GTSRBNet        (reference GTSRB/GTSRBNet.py:8-52)
//   WideResNet-d-w  (reference cifar/wideresnet.py:9-94; 34-10 in the paper's CIFAR attack)
// Every conv / linear layer runs on the fp32 matrix cores through conv_igemm.h; the argmax +
// target-hit count is a wavefront-reduction kernel (utils.py:51-52).
#include "common.h"
#include "conv_igemm.h"
#include "fc_small.h"
#include "conv_halo.h"
#include "conv_halo_cc.h"
#include "conv_halo_bf16.h"
#include "gemm_bf16.h"
#include "conv_halo_cc_bf16.h"
#include "conv21_bf16.h"
#include <stdlib.h>
#include <math.h>
#include <stdarg.h>
#include <string.h>

// ---------------------------------------------------------------------------------------------
// model objects
// ---------------------------------------------------------------------------------------------
struct Layer {
    float* w = nullptr;      // [K][NP]
    float* w3 = nullptr;     // conv2_1 only: [k = ky*16 + kx*3 + ci][NP], row ky*16 + 15 zero (conv_halo.h, PS = 3)
    float* bias = nullptr;   // [NP]
    int Cin = 0, Cout = 0, NP = 0, KH = 1, KW = 1, stride = 1, pad = 0;
    int CinT = 0;            // true (unpadded) input channels, for the algorithmic flop count
    int kchunk = 0;          // 32: k = (ci/32, ky, kx, ci%32) (stride-1 3x3 convs with Cin % 32 == 0), else k = (ky,kx,ci)
    const float* wT = nullptr;   // linear layers of GTSRBNet: the transposed copy [Cout (padded)][ldT], k contiguous (= gb[l].w), for fc_small.h
    int ldT = 0;
    uint16_t* wbf = nullptr; // bf16 copy for conv_halo_bf16.h: [K/16][Cout/32][2][32][8], k = (ky,kx,ci)  (GR_PREC_BF16 only)
};
struct BnDev { float* scale = nullptr; float* shift = nullptr; };
struct WrnBlock {
    int cin, cout, stride;
    BnDev bn1, bn2;
    Layer conv1, conv2, sc;
    Layer conv1T, conv2T, scT;   // transposed (input-gradient pass): flipped taps, Cin <-> Cout, stride 1 (strided layers run on the zero-dilated gradient)
    bool has_sc;
};
struct BagBlock {
    int cin, planes, k, stride;
    Layer c1, c2, c3, ds;        // eval BatchNorm folded into each: w' = w * scale[co], bias' = shift[co]
    bool has_ds;
};
struct gr_model {
    gr_ctx* ctx = nullptr;
    int kind = 0;            // 1 = GTSRBNet, 2 = WideResNet, 3 = PatchGuard (BagNet + masking_defense)
    int precision = GR_PREC_F32;
    int bf16_mask = 0x7f;    // layers that GR_PREC_BF16 runs in bf16: bit 0 = conv2_1 .. bit 5 = conv4_2, bit 6 = fc5 + fc6;
                             // bit 7: keep the activations between them f32 in HBM (they are bf16 when all 7 layers are)
    int ncls = 0;
    std::vector<void*> allocs;
    // GTSRBNet
    float* c11 = nullptr;    // conv1_1: 9 weights + 3 biases
    Layer g[9];              // conv2_1 .. fc7_1
    Layer gb[9];             // the same layers transposed (input-gradient pass): conv with flipped taps / W^T
    // WideResNet
    Layer wconv1, wfc;
    Layer wconv1T, wfcT;     // transposed, for gr_model_input_grad
    std::vector<WrnBlock> blocks;
    BnDev bnf;
    int ch[4] = {0, 0, 0, 0};
    // PatchGuard (patchguard/patchguard_predict.py:21-89): bicubic up-sampling to pg_D, BagNet, robust aggregation
    float* bag_conv1 = nullptr;  // [64][3]
    Layer bag_conv2, bag_fc;
    std::vector<BagBlock> bag;
    int pg_D = 192, pg_h = 0;    // up-sampled size; side of the local-logit map
    size_t pg_big = 0;           // floats of the largest activation per image
    int pg_defense = 1, pg_window = 6;
    float pg_thres = 0.f;
};

static int n_tile(int cout) {   // N tile the dispatcher will use for this many output channels
    if (cout % 160 == 0) return 160;
    if (cout >= 128) return 128;
    if (cout >= 64) return 64;
    return 32;
}

static int upload(gr_model* m, const std::vector<float>& h, float** d) {
    GR_HIP(m->ctx, hipMalloc((void**)d, h.size() * sizeof(float)));
    m->allocs.push_back(*d);
    GR_HIP(m->ctx, hipMemcpy(*d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    return GR_OK;
}

// OIHW -> [k = (ky*KW+kx)*CinP + ci][NP], zero padded in ci (CinP) and co (NP).
// Stride-1 3x3 convs whose Cin is a multiple of 32 are packed (and accumulated, as in the oracle) in channel
// chunks of 32: k = ((ci/32)*KH*KW + ky*KW + kx)*32 + ci%32.
static int make_layer(gr_model* m, Layer* L, const float* w, const float* bias, int Co, int Ci, int KH, int KW, int stride,
                      int pad) {
    int CinP = (Ci + 3) / 4 * 4;
    int bn = n_tile(Co);
    int NP = (Co + bn - 1) / bn * bn;
    const int kchunk = (KH == 3 && KW == 3 && stride == 1 && Ci % 32 == 0) ? 32 : 0;
    std::vector<float> p(((size_t)KH * KW * CinP + 64) * NP, 0.f);   // + 64 zero rows: the halo kernels' B stream reads one stage ahead
    for (int o = 0; o < Co; o++)
        for (int i = 0; i < Ci; i++)
            for (int y = 0; y < KH; y++)
                for (int x = 0; x < KW; x++) {
                    size_t k = kchunk ? ((size_t)(i / kchunk) * KH * KW + y * KW + x) * kchunk + i % kchunk
                                      : (size_t)(y * KW + x) * CinP + i;
                    p[k * NP + o] = w[(((size_t)o * Ci + i) * KH + y) * KW + x];
                }
    L->kchunk = kchunk;
    int rc = upload(m, p, &L->w);
    if (rc) return rc;
    if (bias) {
        std::vector<float> b(NP, 0.f);
        memcpy(b.data(), bias, sizeof(float) * Co);
        rc = upload(m, b, &L->bias);
        if (rc) return rc;
    }
    L->CinT = Ci; L->Cin = CinP; L->Cout = Co; L->NP = NP; L->KH = KH; L->KW = KW; L->stride = stride; L->pad = pad;
    return GR_OK;
}

// conv2_1 for conv_halo.h's packed three-channel rows: k = ky*16 + kx*3 + ci, k = ky*16 + 15 is a zero row.  The
// accumulation chain visits the real (ky, kx, ci) in the same ascending order as the oracle; the zero-weight steps in
// between leave the accumulator unchanged (fmaf(a, 0, acc) == acc for finite a), so the logits stay bit-identical.
static int make_conv21_rows3(gr_model* m, Layer* L, const float* w /* [Co][3][5][5] */, int Co) {
    const int NP = L->NP;
    std::vector<float> p((size_t)(5 * 16 + 64) * NP, 0.f);
    for (int o = 0; o < Co; o++)
        for (int i = 0; i < 3; i++)
            for (int y = 0; y < 5; y++)
                for (int x = 0; x < 5; x++) p[(size_t)(y * 16 + x * 3 + i) * NP + o] = w[(((size_t)o * 3 + i) * 5 + y) * 5 + x];
    return upload(m, p, &L->w3);
}

// float -> bfloat16, round to nearest even (what v_cvt_pk_bf16_f32 does; no NaNs in weights)
static uint16_t bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// bf16 operand image of a 5x5 conv for conv_halo_bf16.h (OIHW f32 in): the 16 bytes a lane (cout lp, k half lh) feeds
// one v_mfma_f32_32x32x16_bf16 are contiguous, lanes and n tiles follow each other, k = (ky*5+kx)*Ci + ci.
static int make_layer_bf16(gr_model* m, Layer* L, const float* w, int Co, int Ci) {
    if (Ci % 16 || Co % 32) return gr_fail(m->ctx, GR_EINVAL, "bf16 conv: Cin %% 16 / Cout %% 32");
    const int K = 25 * Ci, NT32 = Co / 32;
    std::vector<uint16_t> p((size_t)K * Co);
    for (int k = 0; k < K; k++) {
        const int tap = k / Ci, ci = k % Ci, kb = k / 16, lh = (k % 16) / 8, e = k % 8;
        for (int o = 0; o < Co; o++)
            p[((((size_t)kb * NT32 + o / 32) * 2 + lh) * 32 + o % 32) * 8 + e] = bf16_rne(w[((size_t)o * Ci + ci) * 25 + tap]);
    }
    GR_HIP(m->ctx, hipMalloc((void**)&L->wbf, p.size() * sizeof(uint16_t)));
    m->allocs.push_back(L->wbf);
    GR_HIP(m->ctx, hipMemcpy(L->wbf, p.data(), p.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    return GR_OK;
}

// conv2_1 for conv21_bf16.h: k-block = (ky, half of the kernel row), k inside = kx*4 + c, zero where kx > 4 or c == 3
static int make_conv21_bf16(gr_model* m, Layer* L, const float* w /* [32][3][5][5] */) {
    std::vector<uint16_t> p((size_t)10 * 2 * 32 * 8, 0);
    for (int ky = 0; ky < 5; ky++)
        for (int cb = 0; cb < 2; cb++)
            for (int lh = 0; lh < 2; lh++)
                for (int e = 0; e < 8; e++) {
                    const int rowk = cb * 16 + lh * 8 + e, kx = rowk / 4, ch = rowk % 4;
                    if (kx > 4 || ch > 2) continue;
                    for (int o = 0; o < 32; o++)
                        p[((((size_t)(ky * 2 + cb)) * 2 + lh) * 32 + o) * 8 + e] = bf16_rne(w[((size_t)o * 3 + ch) * 25 + ky * 5 + kx]);
                }
    GR_HIP(m->ctx, hipMalloc((void**)&L->wbf, p.size() * sizeof(uint16_t)));
    m->allocs.push_back(L->wbf);
    GR_HIP(m->ctx, hipMemcpy(L->wbf, p.data(), p.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    return GR_OK;
}

// WideResNet's stride-1 3x3 convs for conv_halo_cc_bf16.h: k = ((ci/32)*9 + ky*3+kx)*32 + ci%32 (OIHW f32 in)
static int make_layer_bf16_cc(gr_model* m, Layer* L, const float* w, int Co, int Ci) {
    if (Ci % 32 || Co % 32) return gr_fail(m->ctx, GR_EINVAL, "bf16 3x3 conv: Cin %% 32 / Cout %% 32");
    const int NT32 = Co / 32;
    std::vector<uint16_t> p((size_t)9 * Ci * Co);
    for (int ci = 0; ci < Ci; ci++)
        for (int tap = 0; tap < 9; tap++) {
            const int k = ((ci / 32) * 9 + tap) * 32 + ci % 32, kb = k / 16, lh = (k % 16) / 8, e = k % 8;
            for (int o = 0; o < Co; o++)
                p[((((size_t)kb * NT32 + o / 32) * 2 + lh) * 32 + o % 32) * 8 + e] = bf16_rne(w[((size_t)o * Ci + ci) * 9 + tap]);
        }
    GR_HIP(m->ctx, hipMalloc((void**)&L->wbf, p.size() * sizeof(uint16_t)));
    m->allocs.push_back(L->wbf);
    GR_HIP(m->ctx, hipMemcpy(L->wbf, p.data(), p.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    return GR_OK;
}

// the same operand image for a linear layer, w = [N][K] row-major (PyTorch [out, in])
static int make_fc_bf16(gr_model* m, Layer* L, const float* w, int N, int K) {
    if (K % 16 || N % 32) return gr_fail(m->ctx, GR_EINVAL, "bf16 linear: K %% 16 / N %% 32");
    const int NT32 = N / 32;
    std::vector<uint16_t> p((size_t)K * N);
    for (int n = 0; n < N; n++)
        for (int k = 0; k < K; k++)
            p[((((size_t)(k / 16) * NT32 + n / 32) * 2 + (k % 16) / 8) * 32 + n % 32) * 8 + k % 8] = bf16_rne(w[(size_t)n * K + k]);
    GR_HIP(m->ctx, hipMalloc((void**)&L->wbf, p.size() * sizeof(uint16_t)));
    m->allocs.push_back(L->wbf);
    GR_HIP(m->ctx, hipMemcpy(L->wbf, p.data(), p.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    return GR_OK;
}

// Transposed conv for the input-gradient pass: dX = conv_stride1(dY (zero-dilated if the layer is strided), W') with
// W'[ci][co][ky][kx] = W[co][ci][KH-1-ky][KW-1-kx], same padding.  w is OIHW.
static int make_layer_T(gr_model* m, Layer* L, const float* w, int Co, int Ci, int KH, int KW, int pad) {
    std::vector<float> wt((size_t)Co * Ci * KH * KW);
    const int kk = KH * KW;
    for (int o = 0; o < Co; o++)
        for (int i = 0; i < Ci; i++)
            for (int k = 0; k < kk; k++) wt[((size_t)i * Co + o) * kk + (kk - 1 - k)] = w[((size_t)o * Ci + i) * kk + k];
    return make_layer(m, L, wt.data(), nullptr, Ci, Co, KH, KW, 1, pad);
}

// eval BatchNorm folded as PyTorch's CPU kernel does: alpha = w/sqrt(var+eps), beta = b - mean*alpha
static int make_bn(gr_model* m, BnDev* bn, const float* w, const float* b, const float* mean, const float* var, int C) {
    int NP = (C + n_tile(C) - 1) / n_tile(C) * n_tile(C);
    std::vector<float> sc(NP, 0.f), sh(NP, 0.f);
    for (int c = 0; c < C; c++) {
        float invstd = 1.0f / sqrtf(var[c] + 1e-5f);
        sc[c] = w[c] * invstd;
        sh[c] = b[c] - mean[c] * sc[c];
    }
    int rc = upload(m, sc, &bn->scale);
    if (rc) return rc;
    return upload(m, sh, &bn->shift);
}

static int expect(gr_ctx* ctx, const int64_t* numel, int idx, int64_t want) {
    if (numel[idx] != want) return gr_fail(ctx, GR_EINVAL, "tensor %d has %lld elements, expected %lld", idx,
                                           (long long)numel[idx], (long long)want);
    return GR_OK;
}

extern "C" int gr_model_load_gtsrb(gr_ctx* ctx, const float* const* t, const int64_t* numel, int n, gr_model** out) {
    if (!ctx) return GR_EINVAL;
    if (!t || !numel || !out || n != 20) return gr_fail(ctx, GR_EINVAL, "gr_model_load_gtsrb: need 20 tensors");
    GR_HIP(ctx, hipSetDevice(ctx->device));
    static const int co[10] = {3, 32, 32, 64, 64, 128, 128, 1024, 1024, 43};
    static const int ci[10] = {3, 3, 32, 32, 64, 64, 128, 2048, 1024, 1024};
    static const int ks[10] = {1, 5, 5, 5, 5, 5, 5, 1, 1, 1};
    for (int l = 0; l < 10; l++) {
        int rc = expect(ctx, numel, 2 * l, (int64_t)co[l] * ci[l] * ks[l] * ks[l]);
        if (rc) return rc;
        rc = expect(ctx, numel, 2 * l + 1, co[l]);
        if (rc) return rc;
        if (!t[2 * l] || !t[2 * l + 1]) return gr_fail(ctx, GR_EINVAL, "null tensor %d", 2 * l);
    }
    gr_model* m = new gr_model();
    m->ctx = ctx; m->kind = 1; m->ncls = 43;
    int rc;
    {
        std::vector<float> c(12);
        memcpy(c.data(), t[0], 9 * sizeof(float));
        memcpy(c.data() + 9, t[1], 3 * sizeof(float));
        rc = upload(m, c, &m->c11);
    }
    for (int l = 1; l <= 6 && !rc; l++) rc = make_layer(m, &m->g[l - 1], t[2 * l], t[2 * l + 1], co[l], ci[l], 5, 5, 1, 2);
    if (!rc) rc = make_conv21_rows3(m, &m->g[0], t[2], co[1]);
    if (!rc) rc = make_conv21_bf16(m, &m->g[0], t[2]);
    for (int l = 2; l <= 6 && !rc; l++) rc = make_layer_bf16(m, &m->g[l - 1], t[2 * l], co[l], ci[l]);   // conv2_2 .. conv4_2
    if (!rc) {
        // fc5_1: the reference flattens NCHW (k = c*16 + y*4 + x, GTSRBNet.py:47); activations here are
        // NHWC, so the columns are permuted to k' = (y*4+x)*128 + c.
        std::vector<float> w5((size_t)1024 * 2048);
        for (int o = 0; o < 1024; o++)
            for (int c = 0; c < 128; c++)
                for (int s = 0; s < 16; s++) w5[(size_t)o * 2048 + s * 128 + c] = t[14][(size_t)o * 2048 + c * 16 + s];
        rc = make_layer(m, &m->g[6], w5.data(), t[15], 1024, 2048, 1, 1, 1, 0);
        if (!rc) rc = make_fc_bf16(m, &m->g[6], w5.data(), 1024, 2048);
    }
    if (!rc) rc = make_layer(m, &m->g[7], t[16], t[17], 1024, 1024, 1, 1, 1, 0);
    if (!rc) rc = make_fc_bf16(m, &m->g[7], t[16], 1024, 1024);
    if (!rc) rc = make_layer(m, &m->g[8], t[18], t[19], 43, 1024, 1, 1, 1, 0);
    // transposed layers for gr_model_input_grad: dX = conv(dY, W') with W'[ci][co][ky][kx] = W[co][ci][4-ky][4-kx]
    for (int l = 1; l <= 6 && !rc; l++) {
        std::vector<float> wt((size_t)co[l] * ci[l] * 25);
        for (int o = 0; o < co[l]; o++)
            for (int i = 0; i < ci[l]; i++)
                for (int k = 0; k < 25; k++) wt[((size_t)i * co[l] + o) * 25 + (24 - k)] = t[2 * l][((size_t)o * ci[l] + i) * 25 + k];
        rc = make_layer(m, &m->gb[l - 1], wt.data(), nullptr, ci[l], co[l], 5, 5, 1, 2);
    }
    if (!rc) {      // fc5^T: output index in NHWC-flatten order, as the activations are laid out
        std::vector<float> wt((size_t)2048 * 1024);
        for (int o = 0; o < 1024; o++)
            for (int c = 0; c < 128; c++)
                for (int sp = 0; sp < 16; sp++) wt[(size_t)(sp * 128 + c) * 1024 + o] = t[14][(size_t)o * 2048 + c * 16 + sp];
        rc = make_layer(m, &m->gb[6], wt.data(), nullptr, 2048, 1024, 1, 1, 1, 0);
    }
    if (!rc) {
        std::vector<float> wt((size_t)1024 * 1024);
        for (int o = 0; o < 1024; o++)
            for (int i = 0; i < 1024; i++) wt[(size_t)i * 1024 + o] = t[16][(size_t)o * 1024 + i];
        rc = make_layer(m, &m->gb[7], wt.data(), nullptr, 1024, 1024, 1, 1, 1, 0);
    }
    if (!rc) {      // fc7^T with the 43 logit gradients padded to 64 input channels
        std::vector<float> wt((size_t)1024 * 64, 0.f);
        for (int o = 0; o < 43; o++)
            for (int i = 0; i < 1024; i++) wt[(size_t)i * 64 + o] = t[18][(size_t)o * 1024 + i];
        rc = make_layer(m, &m->gb[8], wt.data(), nullptr, 1024, 64, 1, 1, 1, 0);
    }
    if (rc) { gr_model_destroy(m); return rc; }
    // small launches read the linear layers' weights k-contiguous (fc_small.h): that is the transposed layer's [k = co][n = ci]
    for (int l = 6; l <= 8; l++) { m->g[l].wT = m->gb[l].w; m->g[l].ldT = m->gb[l].NP; }
    *out = m;
    return GR_OK;
}

extern "C" int gr_model_load_wrn(gr_ctx* ctx, const float* const* t, const int64_t* numel, int n, int depth, int widen,
                                 int ncls, gr_model** out) {
    if (!ctx) return GR_EINVAL;
    if (!t || !numel || !out || depth < 10 || (depth - 4) % 6 != 0 || widen < 1 || ncls < 1 || ncls > 32)
        return gr_fail(ctx, GR_EINVAL, "gr_model_load_wrn: bad arguments");
    GR_HIP(ctx, hipSetDevice(ctx->device));
    int nb = (depth - 4) / 6;
    int ch[4] = {16, 16 * widen, 32 * widen, 64 * widen};
    // count + size check first
    int want = 1;
    for (int g = 0; g < 3; g++)
        for (int i = 0; i < nb; i++) want += 10 + (((i == 0 ? ch[g] : ch[g + 1]) != ch[g + 1]) ? 1 : 0);
    want += 6;
    if (n != want) return gr_fail(ctx, GR_EINVAL, "gr_model_load_wrn: %d tensors given, %d expected", n, want);
    for (int i = 0; i < n; i++)
        if (!t[i]) return gr_fail(ctx, GR_EINVAL, "null tensor %d", i);
    gr_model* m = new gr_model();
    m->ctx = ctx; m->kind = 2; m->ncls = ncls;
    memcpy(m->ch, ch, sizeof(ch));
    int ti = 0, rc = expect(ctx, numel, 0, 16 * 3 * 9);
    if (!rc) rc = make_layer(m, &m->wconv1, t[ti], nullptr, 16, 3, 3, 3, 1, 1);
    if (!rc) rc = make_layer_T(m, &m->wconv1T, t[ti], 16, 3, 3, 3, 1);
    ti++;
    for (int g = 0; g < 3 && !rc; g++)
        for (int i = 0; i < nb && !rc; i++) {
            WrnBlock b;
            b.cin = i == 0 ? ch[g] : ch[g + 1];
            b.cout = ch[g + 1];
            b.stride = (i == 0 && g > 0) ? 2 : 1;
            b.has_sc = b.cin != b.cout;
            for (int q = 0; q < 4 && !rc; q++) rc = expect(ctx, numel, ti + q, b.cin);
            if (!rc) rc = make_bn(m, &b.bn1, t[ti], t[ti + 1], t[ti + 2], t[ti + 3], b.cin);
            ti += 4;
            if (!rc) rc = expect(ctx, numel, ti, (int64_t)b.cout * b.cin * 9);
            if (!rc) rc = make_layer(m, &b.conv1, t[ti], nullptr, b.cout, b.cin, 3, 3, b.stride, 1);
            if (!rc && b.cin % 32 == 0 && b.cout % 160 == 0) rc = make_layer_bf16_cc(m, &b.conv1, t[ti], b.cout, b.cin);   // stride 1 and 2
            if (!rc) rc = make_layer_T(m, &b.conv1T, t[ti], b.cout, b.cin, 3, 3, 1);
            ti++;
            for (int q = 0; q < 4 && !rc; q++) rc = expect(ctx, numel, ti + q, b.cout);
            if (!rc) rc = make_bn(m, &b.bn2, t[ti], t[ti + 1], t[ti + 2], t[ti + 3], b.cout);
            ti += 4;
            if (!rc) rc = expect(ctx, numel, ti, (int64_t)b.cout * b.cout * 9);
            if (!rc) rc = make_layer(m, &b.conv2, t[ti], nullptr, b.cout, b.cout, 3, 3, 1, 1);
            if (!rc && b.cout % 160 == 0) rc = make_layer_bf16_cc(m, &b.conv2, t[ti], b.cout, b.cout);
            if (!rc) rc = make_layer_T(m, &b.conv2T, t[ti], b.cout, b.cout, 3, 3, 1);
            ti++;
            if (b.has_sc) {
                if (!rc) rc = expect(ctx, numel, ti, (int64_t)b.cout * b.cin);
                if (!rc) rc = make_layer(m, &b.sc, t[ti], nullptr, b.cout, b.cin, 1, 1, b.stride, 0);
                if (!rc) rc = make_layer_T(m, &b.scT, t[ti], b.cout, b.cin, 1, 1, 0);
                ti++;
            }
            m->blocks.push_back(b);
        }
    for (int q = 0; q < 4 && !rc; q++) rc = expect(ctx, numel, ti + q, ch[3]);
    if (!rc) rc = make_bn(m, &m->bnf, t[ti], t[ti + 1], t[ti + 2], t[ti + 3], ch[3]);
    ti += 4;
    if (!rc) rc = expect(ctx, numel, ti, (int64_t)ncls * ch[3]);
    if (!rc) rc = expect(ctx, numel, ti + 1, ncls);
    if (!rc) rc = make_layer(m, &m->wfc, t[ti], t[ti + 1], ncls, ch[3], 1, 1, 1, 0);
    if (!rc) {      // fc^T with the logit gradients padded to 64 input channels (k_ce_grad's row width)
        std::vector<float> wt((size_t)ch[3] * 64, 0.f);
        for (int o = 0; o < ncls; o++)
            for (int i = 0; i < ch[3]; i++) wt[(size_t)i * 64 + o] = t[ti][(size_t)o * ch[3] + i];
        rc = make_layer(m, &m->wfcT, wt.data(), nullptr, ch[3], 64, 1, 1, 1, 0);
    }
    if (rc) { gr_model_destroy(m); return rc; }
    *out = m;
    return GR_OK;
}

// ---- PatchGuard: BagNet (inspire-group/PatchGuard nets/bagnet.py = Brendel & Bethge's bagnets/pytorchnet.py; third-party
// source absent from the reference, architecture restated -- oracle/graphite_oracle.c or_bagnet_forward has the text) ----
// conv (no bias) followed by eval BatchNorm, folded: w' = w * scale[co] (float32), bias' = shift[co]
static int make_layer_bn(gr_model* m, Layer* L, const float* w, const float* const* bn, int Co, int Ci, int K, int stride) {
    std::vector<float> ws((size_t)Co * Ci * K * K), sh(Co);
    for (int o = 0; o < Co; o++) {
        const float invstd = 1.0f / sqrtf(bn[3][o] + 1e-5f);
        const float sc = bn[0][o] * invstd;
        sh[o] = bn[1][o] - bn[2][o] * sc;
        for (size_t e = 0; e < (size_t)Ci * K * K; e++) ws[(size_t)o * Ci * K * K + e] = w[(size_t)o * Ci * K * K + e] * sc;
    }
    int rc = make_layer(m, L, ws.data(), sh.data(), Co, Ci, K, K, stride, 0);
    // bf16 throughput mode: stride-1 1x1 convs are plain GEMMs over the NHWC rows (gemm_bf16.h); the FOLDED weights are rounded
    if (!rc && K == 1 && stride == 1 && Ci % 32 == 0 && Co % 64 == 0) rc = make_fc_bf16(m, L, ws.data(), Co, Ci);
    return rc;
}

extern "C" int gr_model_load_bagnet(gr_ctx* ctx, const float* const* t, const int64_t* numel, int n, const int* layers,
                                    const int* strides, const int* kernel3, int ncls, int upsample, gr_model** out) {
    if (!ctx) return GR_EINVAL;
    if (!t || !numel || !out || !layers || !strides || !kernel3 || ncls < 1 || ncls > 256 || upsample < 8 || upsample > 1024)
        return gr_fail(ctx, GR_EINVAL, "gr_model_load_bagnet: bad arguments");
    int want = 6;
    for (int g = 0; g < 4; g++) {
        if (layers[g] < 1 || layers[g] > 64 || strides[g] < 1 || strides[g] > 2 || kernel3[g] < 0 || kernel3[g] > 1)
            return gr_fail(ctx, GR_EINVAL, "gr_model_load_bagnet: stage %d: layers %d, stride %d, kernel3 %d unsupported", g,
                           layers[g], strides[g], kernel3[g]);
        want += 15 * layers[g] + 5;
    }
    want += 2;
    if (n != want) return gr_fail(ctx, GR_EINVAL, "gr_model_load_bagnet: %d tensors given, %d expected", n, want);
    for (int i = 0; i < n; i++)
        if (!t[i]) return gr_fail(ctx, GR_EINVAL, "null tensor %d", i);
    GR_HIP(ctx, hipSetDevice(ctx->device));
    gr_model* m = new gr_model();
    m->ctx = ctx; m->kind = 3; m->ncls = ncls; m->pg_D = upsample;
    int ti = 0, rc = expect(ctx, numel, 0, 64 * 3);
    if (!rc) { std::vector<float> c(t[0], t[0] + 192); rc = upload(m, c, &m->bag_conv1); }
    ti = 1;
    if (!rc) rc = expect(ctx, numel, ti, 64 * 64 * 9);
    for (int q = 1; q <= 4 && !rc; q++) rc = expect(ctx, numel, ti + q, 64);
    if (!rc) rc = make_layer_bn(m, &m->bag_conv2, t[ti], t + ti + 1, 64, 64, 3, 1);
    ti += 5;
    int inpl = 64, H = upsample - 2;
    size_t big = (size_t)upsample * upsample * 64;
    for (int g = 0; g < 4 && !rc; g++)
        for (int i = 0; i < layers[g] && !rc; i++) {
            BagBlock b;
            b.planes = 64 << g; b.cin = inpl; b.k = kernel3[g] > i ? 3 : 1; b.stride = i == 0 ? strides[g] : 1; b.has_ds = i == 0;
            const int co[3] = {b.planes, b.planes, 4 * b.planes}, ci[3] = {b.cin, b.planes, b.planes}, kk[3] = {1, b.k, 1};
            Layer* L[3] = {&b.c1, &b.c2, &b.c3};
            for (int j = 0; j < 3 && !rc; j++) {
                rc = expect(ctx, numel, ti, (int64_t)co[j] * ci[j] * kk[j] * kk[j]);
                for (int q = 1; q <= 4 && !rc; q++) rc = expect(ctx, numel, ti + q, co[j]);
                if (!rc) rc = make_layer_bn(m, L[j], t[ti], t + ti + 1, co[j], ci[j], kk[j], j == 1 ? b.stride : 1);
                ti += 5;
            }
            if (b.has_ds && !rc) {
                rc = expect(ctx, numel, ti, (int64_t)4 * b.planes * b.cin);
                for (int q = 1; q <= 4 && !rc; q++) rc = expect(ctx, numel, ti + q, 4 * b.planes);
                if (!rc) rc = make_layer_bn(m, &b.ds, t[ti], t + ti + 1, 4 * b.planes, b.cin, 1, b.stride);
                ti += 5;
            }
            if ((size_t)H * H * b.cin > big) big = (size_t)H * H * b.cin;
            if (H < b.k) rc = gr_fail(ctx, GR_EINVAL, "gr_model_load_bagnet: a %d x %d input is too small for this net", upsample, upsample);
            H = (H - b.k) / b.stride + 1;
            if ((size_t)H * H * 4 * b.planes > big) big = (size_t)H * H * 4 * b.planes;
            inpl = 4 * b.planes;
            m->bag.push_back(b);
        }
    if (!rc) rc = expect(ctx, numel, ti, (int64_t)ncls * inpl);
    if (!rc) rc = expect(ctx, numel, ti + 1, ncls);
    if (!rc) rc = make_layer(m, &m->bag_fc, t[ti], t[ti + 1], ncls, inpl, 1, 1, 1, 0);
    if (!rc && H < 1) rc = gr_fail(ctx, GR_EINVAL, "gr_model_load_bagnet: empty local-logit map");
    m->pg_h = H; m->pg_big = big;
    m->pg_window = H < 6 ? H : 6;            // ceil((30 + 17 - 1) / 8), patchguard_predict.py:33-36
    if (rc) { gr_model_destroy(m); return rc; }
    *out = m;
    return GR_OK;
}

extern "C" int gr_pg_configure(gr_model* m, int defense, int window, float thres) {
    if (!m) return GR_EINVAL;
    if (m->kind != 3) return gr_fail(m->ctx, GR_EINVAL, "gr_pg_configure: not a PatchGuard model");
    if (window < 1 || window > m->pg_h) return gr_fail(m->ctx, GR_EINVAL, "gr_pg_configure: window %d outside 1..%d", window, m->pg_h);
    m->pg_defense = defense ? 1 : 0; m->pg_window = window; m->pg_thres = thres;
    return GR_OK;
}
int net_pg_out_size(const gr_model* m) { return (m && m->kind == 3) ? m->pg_h : 0; }
extern "C" int gr_pg_out_size(const gr_model* m) { return net_pg_out_size(m); }

extern "C" void gr_model_destroy(gr_model* m) {
    if (!m) return;
    for (void* p : m->allocs) (void)hipFree(p);
    delete m;
}
extern "C" int gr_model_num_classes(const gr_model* m) { return m ? m->ncls : GR_EINVAL; }

extern "C" int gr_model_set_precision(gr_model* m, int precision) {
    if (!m) return GR_EINVAL;
    if (precision != GR_PREC_F32 && precision != GR_PREC_BF16) return gr_fail(m->ctx, GR_EINVAL, "gr_model_set_precision: %d", precision);
    if (precision == GR_PREC_BF16 && m->kind == 2) {
        bool any = false;
        for (const WrnBlock& b : m->blocks) any = any || b.conv1.wbf || b.conv2.wbf;
        if (!any) return gr_fail(m->ctx, GR_EINVAL, "gr_model_set_precision: this WideResNet has no layer the bf16 kernels cover "
                                                     "(stride-1 3x3 convs with Cout %% 160 == 0)");
    }
    m->precision = precision;
    return GR_OK;
}
extern "C" int gr_model_get_precision(const gr_model* m) { return m ? m->precision : GR_EINVAL; }
extern "C" int gr_model_set_bf16_layers(gr_model* m, unsigned mask) {
    if (!m) return GR_EINVAL;
    if (mask & ~0xffu) return gr_fail(m->ctx, GR_EINVAL, "gr_model_set_bf16_layers: mask 0x%x has bits outside 0xff", mask);
    m->bf16_mask = (int)mask;
    return GR_OK;
}

// ---------------------------------------------------------------------------------------------
// small kernels
// ---------------------------------------------------------------------------------------------
// GTSRBNet conv1_1 (1x1, 3->3) + ReLU fused with the CHW -> NHWC4 layout change (GTSRBNet.py:36).
// c11 = 9 weights [o][i] + 3 biases.  WITH_CONV=false: layout change only (WideResNet input).
template <bool WITH_CONV>
__global__ void k_input_nhwc4(const float* __restrict__ x, const float* __restrict__ c11, int N, float4* __restrict__ out) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * 1024) return;
    int n = idx >> 10, p = idx & 1023;
    const float* xi = x + (size_t)n * 3072 + p;
    float v0 = xi[0], v1 = xi[1024], v2 = xi[2048];
    float4 o;
    if (WITH_CONV) {
        float r[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            float a = 0.f;
            a = fmaf(v0, c11[k * 3 + 0], a);
            a = fmaf(v1, c11[k * 3 + 1], a);
            a = fmaf(v2, c11[k * 3 + 2], a);
            a = a + c11[9 + k];
            r[k] = a > 0.f ? a : 0.f;
        }
        o = make_float4(r[0], r[1], r[2], 0.f);
    } else {
        o = make_float4(v0, v1, v2, 0.f);
    }
    out[idx] = o;
}

// avg_pool2d(8) over an [N, P, C] NHWC tensor -> [N, C]; sequential sum over pixels, times 1/P
__global__ void k_avgpool(const float* __restrict__ in, int N, int P, int C, float* __restrict__ out) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * C) return;
    int n = idx / C, c = idx % C;
    const float* s = in + (size_t)n * P * C + c;
    float sum = 0.f;
    for (int p = 0; p < P; p++) sum += s[(size_t)p * C];
    out[idx] = sum * (1.0f / (float)P);
}

// torch.max(out,1): first maximum.  One lane per image; hits reduced across the wavefront when all
// 64 lanes belong to one candidate (ballot + popcount -> one atomic), per-lane atomics otherwise.
__global__ void k_argmax_hits(const float* __restrict__ logits, int N, int ncls, long long n_begin, int T, int target,
                              int* __restrict__ hits, long long* __restrict__ preds, int tie_last) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    bool live = idx < N;
    int best = 0;
    if (live) {
        const float* l = logits + (size_t)idx * ncls;
        float bv = l[0];
        for (int c = 1; c < ncls; c++) {
            float v = l[c];
            if (tie_last ? v >= bv : v > bv) { bv = v; best = c; }
        }
        if (preds) preds[idx] = best;
    }
    if (!hits) return;
    bool hit = live && best == target;
    long long n = n_begin + idx;
    int cand = live ? (int)(n / T) : -1;
    int lane = threadIdx.x & 63;
    int c0 = __shfl(cand, 0), c63 = __shfl(cand, 63);
    unsigned long long bal = __ballot(hit);
    if (c0 == c63 && c0 >= 0) {
        if (lane == 0 && bal) atomicAdd(&hits[c0], __popcll(bal));
    } else if (hit) {
        atomicAdd(&hits[cand], 1);
    }
}

// ---------------------------------------------------------------------------------------------
// layer dispatch
// ---------------------------------------------------------------------------------------------
static bool use_igemm_convs();
static bool small_launch_ok();
static int fc16_max_rows();
static bool prof_shapes() {          // GRAPHITE_PROF_SHAPES=1: profile entries of the igemm kernel carry the GEMM shape (tuning runs)
    static int v = -1;
    if (v < 0) { const char* e = getenv("GRAPHITE_PROF_SHAPES"); v = (e && e[0] == '1') ? 1 : 0; }
    return v == 1;
}
static int halo_variant(int layer);
static int bf16_variant(int layer);

// does this WideResNet conv launch go to conv_halo_cc_bf16.h in the bf16 mode?
static bool cc_bf16_ok(const Layer& L, int H, int precision) {
    return precision == GR_PREC_BF16 && L.wbf && L.KH == 3 && L.KW == 3 && L.pad == 1 && (L.stride == 1 || L.stride == 2) &&
           L.Cin % 32 == 0 && L.Cout % 160 == 0 && !L.bias && (H == 32 || H == 16 || (H == 8 && L.stride == 1));
}

// in_bf / act_bf (bf16 kernel only): `in` holds bf16, `out_act` is written as bf16
static int run_conv(gr_ctx* ctx, const Layer& L, const float* in, int B, int H, int W, const float* res, const BnDev* act,
                    float* out_raw, float* out_act, bool relu, bool pool, hipStream_t st, const float* gate = nullptr,
                    int precision = GR_PREC_F32, bool in_bf = false, bool act_bf = false, int out_hw = 0) {
    ConvParams p;
    p.in = in; p.wgt = L.w; p.bias = L.bias; p.res = res; p.gate = gate;
    p.scale = act ? act->scale : nullptr; p.shift = act ? act->shift : nullptr;
    p.out_raw = out_raw; p.out_act = out_act;
    p.H = H; p.W = W; p.Cin = L.Cin;
    p.Ho = (H + 2 * L.pad - L.KH) / L.stride + 1;
    p.Wo = (W + 2 * L.pad - L.KW) / L.stride + 1;
    if (out_hw > 0) {               // only the top-left out_hw x out_hw outputs (BagNet crops the residual branch to that corner)
        if (out_hw > p.Ho || out_hw > p.Wo) return gr_fail(ctx, GR_EINVAL, "conv: crop %d exceeds the output %dx%d", out_hw, p.Ho, p.Wo);
        p.Ho = p.Wo = out_hw;
    }
    p.Cout = L.Cout; p.NP = L.NP; p.KH = L.KH; p.KW = L.KW; p.stride = L.stride; p.pad = L.pad;
    long long M = (long long)B * p.Ho * p.Wo;
    if (M > 0x7fffffff - 512) return gr_fail(ctx, GR_EINVAL, "conv: M too large");
    p.M = (int)M; p.relu = relu ? 1 : 0; p.kchunk = L.kchunk;
    const int K = L.KH * L.KW * L.Cin;
    const int bn = n_tile(L.Cout);
    int bk = 0;
    if (L.Cin % 32 == 0) bk = 32;
    else if (L.Cin % 16 == 0) bk = 16;
    else if (L.Cin == 4 && L.KW == 5) bk = 20;
    else if (L.Cin == 4 && L.KW == 3) bk = 12;
    if (!bk || K % bk) return gr_fail(ctx, GR_EINVAL, "conv: unsupported Cin=%d K=%d", L.Cin, K);
    if (pool && ((p.Ho | p.Wo) & 1)) return gr_fail(ctx, GR_EINVAL, "conv: pooled output must be even");
    hipError_t e = hipErrorInvalidValue;
    // algorithmic work of this launch: 2*M*Cout*KH*KW*Cin flops; bytes = input + weights + outputs once
    const double flops = 2.0 * (double)M * L.Cout * L.KH * L.KW * L.CinT;
    const double bytes = 4.0 * ((double)B * H * W * L.CinT + (double)K * L.Cout +
                                (double)(pool ? M / 4 : M) * L.Cout * ((out_raw ? 1 : 0) + (out_act ? 1 : 0) + (res ? 1 : 0)));
    char pname[64];
    if (L.kchunk && bk != 32) return gr_fail(ctx, GR_ESTATE, "conv: chunked layer needs BK = 32");
    if ((in_bf || act_bf) && !(cc_bf16_ok(L, H, precision) && H == W)) return gr_fail(ctx, GR_ESTATE, "conv: bf16 activations need the bf16 kernel");
    if (cc_bf16_ok(L, H, precision) && !pool && !relu && !gate && H == W) {
        HaloCcBfParams q;
        q.in = in; q.wgt = L.wbf; q.res = res; q.scale = p.scale; q.shift = p.shift; q.out_raw = out_raw; q.out_act = out_act;
        q.B = B; q.H = H; q.W = W; q.Cin = L.Cin; q.Cout = L.Cout;
        const int Hout = H / L.stride;
        snprintf(pname, sizeof(pname), "k_conv_halo_cc_bf16<%dx%d%s>", Hout, Hout, L.stride == 2 ? ",s2" : "");
        ProfScope ps_(ctx, pname, st, flops, bytes);
        //   TH TW NIMG WM TM TN OCC STRIDE; the activation dtypes select one of the instantiations of a shape
#define HCB4(...)                                                                                                  \
        do {                                                                                                       \
            if (in_bf && act_bf) e = launch_conv_halo_cc_bf16<__VA_ARGS__, true, true>(q, st);                     \
            else if (in_bf) e = launch_conv_halo_cc_bf16<__VA_ARGS__, true, false>(q, st);                         \
            else if (act_bf) e = launch_conv_halo_cc_bf16<__VA_ARGS__, false, true>(q, st);                        \
            else e = launch_conv_halo_cc_bf16<__VA_ARGS__, false, false>(q, st);                                   \
        } while (0)
        if (L.stride == 2) {            // 1 wave per SIMD: the slice prefetch of the 17x33 region needs the registers
            if (in_bf) return gr_fail(ctx, GR_ESTATE, "conv: the strided bf16 kernel reads f32 activations");
            if (Hout == 16) { if (act_bf) e = launch_conv_halo_cc_bf16< 8, 16, 1, 4, 1, 5, 1, 2, false, true>(q, st);
                              else e = launch_conv_halo_cc_bf16< 8, 16, 1, 4, 1, 5, 1, 2, false, false>(q, st); }
            else            { if (act_bf) e = launch_conv_halo_cc_bf16< 8,  8, 2, 4, 1, 5, 1, 2, false, true>(q, st);
                              else e = launch_conv_halo_cc_bf16< 8,  8, 2, 4, 1, 5, 1, 2, false, false>(q, st); }
        } else if (H >= 16) HCB4( 8, 16, 1,  4, 1, 5, 2, 1);
        else HCB4( 8,  8, 2,  4, 1, 5, 2, 1);
#undef HCB4
        if (e != hipSuccess) return gr_fail(ctx, GR_EHIP, "conv_halo_cc_bf16 launch (H=%d): %s", H, hipGetErrorString(e));
        return GR_OK;
    }
    if (L.kchunk == 32 && bn == 160 && !pool && !use_igemm_convs() && H == W && (H == 32 || H == 16 || H == 8)) {
        HaloCcParams q;
        q.in = in; q.wgt = L.w; q.bias = L.bias; q.res = res; q.scale = p.scale; q.shift = p.shift;
        q.out_raw = out_raw; q.out_act = out_act;
        q.B = B; q.H = H; q.W = W; q.Cin = L.Cin; q.Cout = L.Cout; q.NP = L.NP; q.relu = p.relu;
        const int v = halo_variant(H == 32 ? 0 : H == 16 ? 1 : 2);
        snprintf(pname, sizeof(pname), "k_conv_halo_cc<%dx%d,v%d>", H, H, v);
        ProfScope ps_(ctx, pname, st, flops, bytes);
        //                                   KS  CC  TH  TW NIMG WM WN TM TN KU  RP
#define HC(...) e = launch_conv_halo_cc<__VA_ARGS__>(q, st)
        if (H >= 16) {
            switch (v) {                  // variant 0 = fastest measured (profiles/r01_conv_tune2.md)
            case 0: HC(3, 32, 16, 16, 1,  4, 1, 2, 5, 4, false, 2); break;
            case 1: HC(3, 32, 16, 16, 1,  4, 1, 2, 5, 2, false, 2); break;
            case 2: HC(3, 32, 16, 16, 1,  4, 1, 2, 5, 4, false, 0); break;
            case 3: HC(3, 32, 16, 16, 1,  4, 1, 2, 5, 4, false, 1); break;
            case 4: HC(3, 32, 16, 16, 1,  4, 1, 2, 5, 4, false, 4); break;
            case 5: HC(3, 32, 16, 16, 1,  4, 1, 2, 5, 4, false, 6); break;
            case 6: HC(3, 32, 16, 16, 1,  8, 1, 1, 5, 4, false, 2); break;
            default: return gr_fail(ctx, GR_EINVAL, "conv_halo_cc: no variant %d", v);
            }
        } else {
            switch (v) {
            case 0: HC(3, 32,  8,  8, 4,  4, 1, 2, 5, 4, true , 2); break;
            case 1: HC(3, 32,  8,  8, 4,  4, 1, 2, 5, 2, true , 2); break;
            case 2: HC(3, 32,  8,  8, 4,  4, 1, 2, 5, 4, true , 0); break;
            case 3: HC(3, 32,  8,  8, 4,  4, 1, 2, 5, 4, true , 1); break;
            case 4: HC(3, 32,  8,  8, 4,  4, 1, 2, 5, 4, true , 4); break;
            case 5: HC(3, 32,  8,  8, 4,  4, 1, 2, 5, 4, true , 6); break;
            case 6: HC(3, 32,  8,  8, 4,  8, 1, 1, 5, 4, true , 2); break;
            default: return gr_fail(ctx, GR_EINVAL, "conv_halo_cc: no variant %d", v);
            }
        }
#undef HC
        if (e != hipSuccess) return gr_fail(ctx, GR_EHIP, "conv_halo_cc launch (H=%d): %s", H, hipGetErrorString(e));
        return GR_OK;
    }
#define GO(BK_, WM_, WN_, TM_, TN_, POOL_)                                                                        \
    do {                                                                                                          \
        if (prof_shapes())                                                                                        \
            snprintf(pname, sizeof(pname), "igemm<%d,%d,%d,%d,%d,%d> M%d K%d N%d k%ds%d", BK_, WM_, WN_, TM_, TN_,     \
                     (int)POOL_, p.M, K, L.Cout, L.KH, L.stride);                                                \
        else                                                                                                      \
        snprintf(pname, sizeof(pname), "k_conv_igemm<%d,%d,%d,%d,%d,%d>", BK_, WM_, WN_, TM_, TN_, (int)POOL_);      \
        ProfScope ps_(ctx, pname, st, flops, bytes);                                                              \
        e = launch_conv_igemm<BK_, WM_, WN_, TM_, TN_, POOL_>(p, st);                                             \
    } while (0)
    // Small launches (the attack's 100-image evaluations: M = 100 .. 1000 rows of a linear layer): a 128x128 tile would
    // leave 8 workgroups walking K with 64 MFMAs per wave and chunk while 248 CUs idle.  128x32 tiles put one 32x32
    // tile on each wave (a quarter of the chain time) and 32 workgroups per 128 rows on the chip.  Same k-ascending
    // chain per output: bit-identical results.
    const bool small_fc = small_launch_ok() && bk == 32 && !pool && L.KH == 1 && L.KW == 1 && H == 1 && W == 1 && L.NP % 32 == 0 && !L.kchunk;
    // <= fc16_max_rows(): one 16x16 tile per wave on v_mfma_f32_16x16x4_f32, a chain a quarter as long again (fc_small.h)
    if (small_fc && p.M <= fc16_max_rows() && (K == 1024 || K == 2048) && L.wT && !res && !gate && !act && out_raw && !out_act &&
        ((uintptr_t)in & 15) == 0) {
        static int fv = -1;              // GRAPHITE_FC16_VARIANT: waves per workgroup / stages in flight (tuning runs)
        if (fv < 0) { const char* ev = getenv("GRAPHITE_FC16_VARIANT"); fv = ev ? atoi(ev) : 0; }
        snprintf(pname, sizeof(pname), "k_fc_small<v%d>", fv);
        ProfScope ps_(ctx, pname, st, flops, bytes);
#define FS(W_, ST_, TM_, TN_) e = K == 1024 ? launch_fc_small<W_, ST_, 1024, TM_, TN_>(p, L.wT, L.ldT, st) \
                                           : launch_fc_small<W_, ST_, 2048, TM_, TN_>(p, L.wT, L.ldT, st)
        // measured (scripts/fc_small_check.py, three layers of one forward): 44 us at 100 rows, 94 at 400, 125 at 600 against
        // 148 us flat on the 32x32x2 tiles; 2 or 4 tiles per wave (fewer operand bytes, longer chains) win nowhere below the
        // cross-over at ~700 rows
        switch (fv) {
        case 0: FS(2, 8, 1, 1); break;
        case 1: FS(2, 8, 1, 2); break;
        case 2: FS(2, 8, 2, 1); break;
        case 3: FS(2, 8, 2, 2); break;
        default: return gr_fail(ctx, GR_EINVAL, "fc_small: no variant %d", fv);
        }
#undef FS
    } else if (small_fc && p.M <= 4096) {
        GO(32, 4, 1, 1, 1, false);
    } else
    {
        // K chunk: 16 wherever the layer allows it.  The kernel holds ~170 registers (3 waves per SIMD at most), and with
        // BK = 32 its LDS buffers (67 KB for the 128 x 128 tile) allow only 2 workgroups per CU; BK = 16 halves them, so a
        // third workgroup covers the others' barriers: fc5 / fc6 111 -> 121 TFLOP/s, BagNet-17 90 -> 101 TFLOP/s
        // (gpurun_out/r3e/igemm_variants*.txt; BK = 8 and 256-wide N tiles measured slower).  Channel-chunked layers
        // (kchunk) and the pooled instantiations keep BK = 32.  GRAPHITE_IGEMM_VARIANT=9: BK = 32 everywhere (A/B runs).
        static int gv = -1;
        if (gv < 0) { const char* e = getenv("GRAPHITE_IGEMM_VARIANT"); gv = e ? atoi(e) : 0; }
        // Only for launches that give every CU several 256-row workgroups: the whitebox gradient's transposed layers and the
        // attack's small evaluations launch a few thousand rows and keep the 128-row BK = 32 tiles (measured: 0.37 vs 0.41 ms).
        const bool few256 = ((long long)(p.M + 255) / 256) * (L.NP / bn) < 768;   // 768: fc5 / fc6 of a 32,000-image chunk (1000 workgroups) take the new tile
        const bool k16 = (bk == 16) || (bk == 32 && !pool && !L.kchunk && gv != 9 && !few256);
        if (bn == 32) {
            if (bk == 32) { if (pool) GO(32, 4, 1, 2, 1, true); else GO(32, 4, 1, 2, 1, false); }   // 32-wide N (fc7, BagNet's fc): BK = 32 measured faster
            else if (bk == 16 && !pool) GO(16, 4, 1, 2, 1, false);
            else if (bk == 20 && !pool) GO(20, 4, 1, 2, 1, false);
            else if (bk == 12 && !pool) GO(12, 4, 1, 2, 1, false);
        } else if (bn == 64) {
            if (k16 && !pool) { if (gv == 1) GO(16, 2, 2, 2, 1, false); else GO(16, 4, 2, 2, 1, false); }
            else if (bk == 32) { if (pool) GO(32, 2, 2, 2, 1, true); else GO(32, 2, 2, 2, 1, false); }
        } else if (bn == 128) {
            if (k16 && !pool) { if (gv == 1) GO(16, 2, 2, 2, 2, false); else GO(16, 4, 2, 2, 2, false); }
            else if (bk == 32) { if (pool) GO(32, 2, 2, 2, 2, true); else GO(32, 2, 2, 2, 2, false); }
        } else if (bn == 160) {
            if (k16 && !pool) GO(16, 4, 1, 1, 5, false);
            else if (bk == 32 && !pool) GO(32, 4, 1, 1, 5, false);
        }
    }
#undef GO
    if (e != hipSuccess) return gr_fail(ctx, GR_EHIP, "conv launch (bn=%d bk=%d pool=%d): %s", bn, bk, (int)pool, hipGetErrorString(e));
    return GR_OK;
}

// per-image workspace in floats
static size_t per_image_floats(const gr_model* m) {
    if (m->kind == 1) return 4096 + 32768 + 8192;                  // x4 + bufA + bufB
    if (m->kind == 3) return 4096 + 4 * m->pg_big + (size_t)m->pg_h * m->pg_h * m->ncls;   // x4 + X, T1, T2, R + local logits
    size_t big = (size_t)1024 * m->ch[1];
    if ((size_t)256 * m->ch[2] > big) big = (size_t)256 * m->ch[2];
    return 4096 + 5 * big;                                          // x4 + x, o, u, s, y
}
size_t net_workspace_bytes(const gr_model* m, int N) { return gr_align(per_image_floats(m) * sizeof(float) * (size_t)N); }

// GRAPHITE_HALO_VARIANT="l:v,l:v" picks a tiling variant per layer (tuning runs); default 0.
// GRAPHITE_BF16_VARIANT: the same for the bf16 kernels.
static void parse_variants(const char* name, int (&tab)[8]) {
    for (int i = 0; i < 8; i++) tab[i] = 0;
    const char* e = getenv(name);
    while (e && *e) {
        int l = -1, v = 0;
        if (sscanf(e, "%d:%d", &l, &v) == 2 && l >= 0 && l < 8 && v >= 0 && v < 8) tab[l] = v;
        e = strchr(e, ',');
        if (e) e++;
    }
}
static int halo_variant(int layer) {
    static int tab[8] = {-1, 0, 0, 0, 0, 0, 0, 0};
    if (tab[0] < 0) parse_variants("GRAPHITE_HALO_VARIANT", tab);
    return tab[layer];
}
static int bf16_variant(int layer) {
    static int tab[8] = {-1, 0, 0, 0, 0, 0, 0, 0};
    if (tab[0] < 0) parse_variants("GRAPHITE_BF16_VARIANT", tab);
    return tab[layer];
}

// GTSRBNet's 5x5 convs through the halo-tile kernel (conv_halo.h).  layer = index into gr_model::g.
static int run_gtsrb_halo(gr_ctx* ctx, const Layer& L, int layer, const float* in, int B, int H, float* out, hipStream_t st) {
    HaloParams p;
    p.in = in; p.wgt = L.w; p.bias = L.bias; p.out = out; p.B = B; p.H = H; p.W = H; p.Cout = L.Cout; p.relu = 1;
    { static int dbg = -1; if (dbg < 0) { const char* e = getenv("GRAPHITE_HALO_DBG"); dbg = e ? atoi(e) : 0; } p.dbg = dbg; }
    const bool pool = layer == 1 || layer == 3 || layer == 5;
    const double M = (double)B * H * H;
    const double flops = 2.0 * M * L.Cout * 25 * L.CinT;
    const double bytes = 4.0 * (M * L.CinT + 25.0 * L.CinT * L.Cout + (pool ? M / 4 : M) * L.Cout);
    char pname[64];
    snprintf(pname, sizeof(pname), "k_conv_halo<conv%d_%d>", 2 + layer / 2, 1 + layer % 2);
    ProfScope ps_(ctx, pname, st, flops, bytes);
    hipError_t e = hipErrorInvalidValue;
    int v = halo_variant(layer);
    // Small launches (one or two 100-transform candidates of an attack): variant 7 = 8 wavefronts with ONE
    // 32x32 tile each, twice as many workgroups -- the MFMA chain a wave walks is a quarter to a half as long, which is
    // what a launch that cannot fill the chip pays for.  Same chains per output: bit-identical.
    // Measured (scripts/small_launch.py): conv3_1 gains up to B = 128, conv4_1 up to B = 256; conv3_2 / conv4_2 do not.
    if (v == 0 && small_launch_ok() && ((layer == 2 && B <= 128) || (layer == 4 && B <= 256))) v = 7;
    //                            CIN CPH KH KW  PS  TH  TW NIMG WM WN TM TN  KU   NP  POOL
#define HL(...) e = launch_conv_halo<__VA_ARGS__>(p, st)
    switch (layer * 8 + v) {            // variant 0 of each layer = the fastest measured (profiles/r01_tune_*.log)
    case 2 * 8 + 7: HL( 32, 32, 5, 5,  33,  8, 16, 1,  4, 2, 1, 1,  8,  64, false, 0); break;   // conv3_1, small launch
    case 4 * 8 + 7: HL( 64, 64, 5, 5,  65,  8,  8, 1,  2, 4, 1, 1, 16, 128, false, 0); break;   // conv4_1, small launch
    case 0 * 8 + 0: p.wgt = L.w3; HL( 16,  4, 5, 1,   3, 16, 32, 1,  4, 1, 4, 1,  8,  32, false, 1); break;   // conv2_1, K = 80
    case 0 * 8 + 1: p.wgt = L.w3; HL( 16,  4, 5, 1,   3, 16, 32, 1,  4, 1, 4, 1,  4,  32, false, 1); break;
    case 0 * 8 + 2: p.wgt = L.w3; HL( 16,  4, 5, 1,   3, 16, 32, 1,  4, 1, 4, 1,  8,  32, false, 0); break;
    case 0 * 8 + 3: p.wgt = L.w3; HL( 16,  4, 5, 1,   3,  8, 32, 1,  4, 1, 2, 1,  8,  32, false, 0); break;
    case 0 * 8 + 4: HL( 20,  4, 5, 1,   4, 16, 32, 1,  4, 1, 4, 1,  5,  32, false, 1); break;   // conv2_1 with the pad channel, K = 100
    case 0 * 8 + 5: HL( 20,  4, 5, 1,   4, 16, 32, 1,  4, 1, 4, 1,  5,  32, false, 0); break;
    case 0 * 8 + 6: HL( 20,  4, 5, 1,   4,  8, 32, 1,  4, 1, 2, 1,  5,  32, false, 0); break;
    case 1 * 8 + 0: HL( 32, 32, 5, 5,  33,  8, 32, 1,  4, 1, 2, 1, 16,  32, true , 0); break;   // conv2_2
    case 1 * 8 + 1: HL( 32, 32, 5, 5,  33,  8, 32, 1,  4, 1, 2, 1,  8,  32, true , 0); break;
    case 1 * 8 + 2: HL( 32, 32, 5, 5,  33, 16, 32, 1,  4, 1, 4, 1,  8,  32, true , 0); break;
    case 1 * 8 + 3: HL( 32, 32, 5, 5,  33, 16, 32, 1,  8, 1, 2, 1,  8,  32, true , 0); break;
    case 1 * 8 + 4: HL( 32, 32, 5, 5,  33,  8, 32, 1,  4, 1, 2, 1, 16,  32, true , 2); break;
    case 1 * 8 + 6: HL( 32, 32, 5, 5,  33,  8, 32, 1,  4, 1, 2, 1, 16,  32, true , 18); break;
    case 2 * 8 + 0: HL( 32, 32, 5, 5,  33, 16, 16, 1,  2, 2, 4, 1,  8,  64, false, 0); break;   // conv3_1
    case 2 * 8 + 1: HL( 32, 32, 5, 5,  33, 16, 16, 1,  2, 2, 4, 1, 16,  64, false, 0); break;
    case 2 * 8 + 2: HL( 32, 32, 5, 5,  33, 16, 16, 1,  4, 1, 2, 2,  8,  64, false, 0); break;
    case 2 * 8 + 4: HL( 32, 32, 5, 5,  33, 16, 16, 1,  2, 2, 4, 1,  8,  64, false, 2); break;
    case 2 * 8 + 6: HL( 32, 32, 5, 5,  33, 16, 16, 1,  2, 2, 4, 1,  8,  64, false, 18); break;
    case 3 * 8 + 0: HL( 64, 64, 5, 5,  65,  8, 16, 1,  2, 2, 2, 1, 16,  64, true , 0); break;   // conv3_2
    case 3 * 8 + 1: HL( 64, 64, 5, 5,  65,  8, 16, 1,  2, 2, 2, 1,  8,  64, true , 0); break;
    case 3 * 8 + 2: HL( 64, 64, 5, 5,  65,  8, 16, 1,  2, 2, 2, 1, 32,  64, true , 0); break;
    case 3 * 8 + 4: HL( 64, 64, 5, 5,  65,  8, 16, 1,  2, 2, 2, 1, 16,  64, true , 2); break;
    case 3 * 8 + 6: HL( 64, 64, 5, 5,  65,  8, 16, 1,  2, 2, 2, 1, 16,  64, true , 18); break;
    case 4 * 8 + 0: HL( 64, 64, 5, 5,  65,  8,  8, 2,  1, 4, 4, 1, 16, 128, false, 0); break;   // conv4_1
    case 4 * 8 + 1: HL( 64, 64, 5, 5,  65,  8,  8, 2,  2, 2, 2, 2,  8, 128, false, 0); break;
    case 4 * 8 + 2: HL( 64, 64, 5, 5,  65,  8,  8, 2,  1, 4, 4, 1,  8, 128, false, 0); break;
    case 4 * 8 + 4: HL( 64, 64, 5, 5,  65,  8,  8, 2,  1, 4, 4, 1, 16, 128, false, 2); break;
    case 4 * 8 + 6: HL( 64, 64, 5, 5,  65,  8,  8, 2,  1, 4, 4, 1, 16, 128, false, 18); break;
    case 5 * 8 + 0: HL(128,128, 5, 5, 129,  8,  8, 1,  1, 4, 2, 1, 16, 128, true , 0); break;   // conv4_2
    case 5 * 8 + 1: HL(128,128, 5, 5, 129,  8,  8, 1,  1, 4, 2, 1,  8, 128, true , 0); break;
    case 5 * 8 + 2: HL(128,128, 5, 5, 129,  8,  8, 1,  1, 4, 2, 1, 32, 128, true , 0); break;
    case 5 * 8 + 4: HL(128,128, 5, 5, 129,  8,  8, 1,  1, 4, 2, 1, 16, 128, true , 2); break;
    case 5 * 8 + 6: HL(128,128, 5, 5, 129,  8,  8, 1,  1, 4, 2, 1, 16, 128, true , 18); break;
    default: return gr_fail(ctx, GR_EINVAL, "conv_halo: no variant %d for layer %d", v, layer);
    }
#undef HL
    if (e != hipSuccess) return gr_fail(ctx, GR_EHIP, "conv_halo launch (layer %d): %s", layer, hipGetErrorString(e));
    return GR_OK;
}

// GTSRBNet's conv2_2 .. conv4_2 on the bf16 matrix cores (conv_halo_bf16.h), f32 activations in and out.
static int run_gtsrb_halo_bf16(gr_ctx* ctx, const Layer& L, int layer, const float* in, int B, int H, float* out, hipStream_t st,
                               bool in_bf = false, bool out_bf = false) {      // activations stored as bf16 (variant 0 only)
    HaloBfParams p;
    p.in = in; p.wgt = L.wbf; p.bias = L.bias; p.out = out; p.B = B; p.H = H; p.W = H; p.relu = 1;
    { static int dbg = -1; if (dbg < 0) { const char* e = getenv("GRAPHITE_BF_DBG"); dbg = e ? atoi(e) : 0; } p.dbg = dbg; }
    const bool pool = layer == 1 || layer == 3 || layer == 5;
    const double M = (double)B * H * H;
    const double flops = 2.0 * M * L.Cout * 25 * L.CinT;
    const double bytes = (in_bf ? 2.0 : 4.0) * M * L.CinT + (out_bf ? 2.0 : 4.0) * (pool ? M / 4 : M) * L.Cout + 2.0 * 25.0 * L.CinT * L.Cout;
    char pname[64];
    snprintf(pname, sizeof(pname), "k_conv_halo_bf16<conv%d_%d>", 2 + layer / 2, 1 + layer % 2);
    ProfScope ps_(ctx, pname, st, flops, bytes);
    hipError_t e = hipErrorInvalidValue;
    const int v = bf16_variant(layer);
    //                                 CIN COUT  TH  TW NIMG WM WN TM TN KBS POOL
#define HB(...) e = launch_conv_halo_bf16<__VA_ARGS__>(p, st)
    if (in_bf && out_bf && v != 0) {    // tuning variants of the production configuration (every link bf16)
#define HBB(...) HB(__VA_ARGS__, 3, true, true)
#define HBP(PD, ...) HB(__VA_ARGS__, 3, true, true, PD)
        switch (layer * 8 + v) {
        case 1 * 8 + 1: HBB( 32,  32,  8, 32, 1,  4, 1, 2, 1,  2, true ); break;
        case 1 * 8 + 2: HBB( 32,  32, 16, 32, 1,  8, 1, 2, 1,  2, true ); break;
        case 1 * 8 + 3: HBB( 32,  32, 16, 32, 1,  4, 1, 4, 1, 10, true ); break;
        case 1 * 8 + 4: HBB( 32,  32,  8, 32, 1,  4, 1, 2, 1, 10, true ); break;
        case 2 * 8 + 1: HBB( 32,  64, 16, 16, 1,  4, 2, 2, 1,  2, false); break;
        case 2 * 8 + 2: HBB( 32,  64, 16, 16, 1,  2, 2, 4, 1, 10, false); break;
        case 2 * 8 + 3: HBB( 32,  64, 16, 16, 1,  2, 1, 4, 2,  2, false); break;
        case 2 * 8 + 4: HBB( 32,  64, 16, 16, 1,  4, 1, 2, 2,  2, false); break;
        case 3 * 8 + 1: HBB( 64,  64, 16, 16, 1,  4, 2, 2, 1, 20, true ); break;
        case 3 * 8 + 2: HBB( 64,  64, 16, 16, 1,  4, 1, 2, 2,  4, true ); break;
        case 3 * 8 + 3: HBB( 64,  64, 16, 16, 1,  2, 2, 4, 1,  4, true ); break;
        case 3 * 8 + 4: HBB( 64,  64, 16, 16, 1,  2, 1, 4, 2,  4, true ); break;
        case 4 * 8 + 1: HBB( 64, 128,  8,  8, 2,  2, 4, 2, 1,  2, false); break;
        case 4 * 8 + 2: HBB( 64, 128,  8,  8, 2,  2, 2, 2, 2,  4, false); break;
        case 4 * 8 + 3: HBB( 64, 128,  8,  8, 2,  1, 4, 4, 1,  4, false); break;
        case 5 * 8 + 1: HBP(2, 128, 128,  8,  8, 2,  2, 4, 2, 1,  4, true ); break;
        case 5 * 8 + 2: HBB(128, 128,  8,  8, 2,  2, 4, 2, 1,  8, true ); break;
        case 5 * 8 + 3: HBB(128, 128,  8,  8, 2,  2, 2, 2, 2,  4, true ); break;
        case 5 * 8 + 4: HBB(128, 128,  8,  8, 2,  1, 4, 4, 1,  4, true ); break;
        case 5 * 8 + 5: HBB(128, 128,  8,  8, 1,  1, 2, 2, 2,  2, true ); break;   // 2 waves x (2 x 2 tiles), 56 KB: 2 workgroups per CU
        case 5 * 8 + 6: HBP(1, 128, 128,  8,  8, 1,  1, 2, 2, 2,  1, true ); break;   // the same with 8 KB weight stages: 3 workgroups per CU
        case 5 * 8 + 7: HBB(128, 128,  8,  8, 1,  2, 2, 1, 2,  2, true ); break;   // 4 waves x (1 x 2 tiles)
        default: return gr_fail(ctx, GR_EINVAL, "conv_halo_bf16: no variant %d for layer %d with bf16 links", v, layer);
        }
#undef HBB
#undef HBP
    } else if (in_bf || out_bf) {       // variant 0 with bf16 activations on either side
#define HB4P(PD, ...)                                                                                              \
        do {                                                                                                       \
            if (in_bf && out_bf) HB(__VA_ARGS__, 3, true, true, PD);                                               \
            else if (in_bf) HB(__VA_ARGS__, 3, true, false, PD);                                                   \
            else HB(__VA_ARGS__, 3, false, true, PD);                                                              \
        } while (0)
#define HB4(...) HB4P(2, __VA_ARGS__)
        switch (layer) {
        case 1: HB4( 32,  32, 16, 32, 1,  8, 1, 2, 1, 10, true ); break;
        case 2: HB4( 32,  64, 16, 16, 1,  2, 2, 4, 1,  2, false); break;
        case 3: HB4( 64,  64, 16, 16, 1,  4, 2, 2, 1,  4, true ); break;
        case 4: HB4( 64, 128,  8,  8, 2,  2, 4, 2, 1,  4, false); break;
        case 5: HB4P(0, 128, 128,  8,  8, 2,  2, 4, 2, 1,  4, true ); break;   // fragment read-ahead measured slower here
        default: return gr_fail(ctx, GR_EINVAL, "conv_halo_bf16: layer %d", layer);
        }
#undef HB4
#undef HB4P
    } else
    switch (layer * 8 + v) {
    // variant 0 = fastest measured (profiles/r01_bf16_tune.md): 8 wavefronts per workgroup wherever the tile allows
    case 1 * 8 + 0: HB( 32,  32, 16, 32, 1,  8, 1, 2, 1, 10, true ); break;    // conv2_2
    case 1 * 8 + 1: HB( 32,  32, 16, 32, 1,  8, 1, 2, 1,  2, true ); break;
    case 1 * 8 + 2: HB( 32,  32, 32, 32, 1, 16, 1, 2, 1, 10, true ); break;
    case 1 * 8 + 3: HB( 32,  32,  8, 32, 1,  4, 1, 2, 1,  2, true ); break;
    case 1 * 8 + 4: HB( 32,  32, 16, 32, 1,  4, 1, 4, 1, 10, true ); break;
    case 2 * 8 + 0: HB( 32,  64, 16, 16, 1,  2, 2, 4, 1,  2, false); break;    // conv3_1
    case 2 * 8 + 1: HB( 32,  64, 16, 16, 1,  4, 2, 2, 1,  2, false); break;
    case 2 * 8 + 2: HB( 32,  64, 16, 16, 2,  4, 2, 4, 1,  2, false); break;
    case 2 * 8 + 3: HB( 32,  64,  8, 16, 1,  2, 2, 2, 1,  2, false); break;
    case 2 * 8 + 4: HB( 32,  64, 16, 16, 1,  2, 2, 4, 1, 10, false); break;
    case 3 * 8 + 0: HB( 64,  64, 16, 16, 1,  4, 2, 2, 1,  4, true ); break;    // conv3_2
    case 3 * 8 + 1: HB( 64,  64, 16, 16, 1,  4, 2, 2, 1,  2, true ); break;
    case 3 * 8 + 3: HB( 64,  64,  8, 16, 1,  2, 2, 2, 1,  4, true ); break;
    case 3 * 8 + 4: HB( 64,  64, 16, 16, 1,  2, 2, 4, 1,  4, true ); break;
    case 4 * 8 + 0: HB( 64, 128,  8,  8, 2,  2, 4, 2, 1,  4, false); break;    // conv4_1
    case 4 * 8 + 1: HB( 64, 128,  8,  8, 2,  2, 4, 2, 1,  2, false); break;
    case 4 * 8 + 2: HB( 64, 128,  8,  8, 4,  4, 4, 2, 1,  4, false); break;
    case 4 * 8 + 3: HB( 64, 128,  8,  8, 2,  2, 2, 2, 2,  4, false); break;
    case 4 * 8 + 4: HB( 64, 128,  8,  8, 4,  4, 2, 2, 2,  4, false); break;
    case 5 * 8 + 0: HB(128, 128,  8,  8, 2,  2, 4, 2, 1,  4, true , 3, false, false, 0); break;    // conv4_2
    case 5 * 8 + 1: HB(128, 128,  8,  8, 2,  2, 4, 2, 1,  2, true ); break;
    case 5 * 8 + 3: HB(128, 128,  8,  8, 1,  1, 4, 2, 1,  4, true ); break;
    case 5 * 8 + 4: HB(128, 128,  8,  8, 2,  2, 2, 2, 2,  4, true ); break;
    default: return gr_fail(ctx, GR_EINVAL, "conv_halo_bf16: no variant %d for layer %d", v, layer);
    }
#undef HB
    if (e != hipSuccess) return gr_fail(ctx, GR_EHIP, "conv_halo_bf16 launch (layer %d): %s", layer, hipGetErrorString(e));
    return GR_OK;
}

// fc5 / fc6 of the bf16 mode (gemm_bf16.h)
static int run_fc_bf16(gr_ctx* ctx, const Layer& L, const char* name, const float* in, int M, float* out, hipStream_t st,
                       bool in_bf = false, bool out_bf = false) {
    GemmBfParams p;
    p.a = in; p.w = L.wbf; p.bias = L.bias; p.c = out; p.M = M; p.N = L.Cout; p.K = L.CinT; p.relu = 1; p.res = nullptr;
    char pname[64];
    snprintf(pname, sizeof(pname), "k_gemm_bf16<%s>", name);
    ProfScope ps_(ctx, pname, st, 2.0 * M * L.Cout * L.CinT, 4.0 * M * (L.CinT + L.Cout) + 2.0 * L.CinT * L.Cout);
    hipError_t e = in_bf ? (out_bf ? launch_gemm_bf16<2, 2, 2, 2, 64, true, true>(p, st) : launch_gemm_bf16<2, 2, 2, 2, 64, true, false>(p, st))
                         : (out_bf ? launch_gemm_bf16<2, 2, 2, 2, 64, false, true>(p, st) : launch_gemm_bf16<2, 2, 2, 2, 64>(p, st));
    if (e != hipSuccess) return gr_fail(ctx, GR_EHIP, "gemm_bf16 launch (%s): %s", name, hipGetErrorString(e));
    return GR_OK;
}

// Every activation between two bf16 layers is stored as bf16 (links 0..6).  Round 1 kept the Cin <= 64 consumers on f32
// inputs because their bf16-input instantiations were 1.5-3x slower; the cause was register spilling (228-308 bytes of
// scratch per lane) when the tile-staging loop had all its loads in one unrolled batch -- fixed in conv_halo_bf16.h (UB),
// after which every link is a gain (profiles/r02_bf16_links.txt: forward 8.10 -> 7.59 ms per 32,000 images).
#define BF16_LINKS_DEFAULT 0x7f
static bool small_launch_ok() {      // GRAPHITE_SMALL=0 turns the small-launch tilings off (A/B runs)
    static int v = -1;
    if (v < 0) { const char* e = getenv("GRAPHITE_SMALL"); v = (e && e[0] == '0') ? 0 : 1; }
    return v == 1;
}
static int fc16_max_rows() {         // GRAPHITE_FC16_ROWS: largest linear-layer launch (rows) on the 16x16x4 kernel; 0 turns it off
    static int v = -1;
    if (v < 0) { const char* e = getenv("GRAPHITE_FC16_ROWS"); v = e ? atoi(e) : 640; if (v < 0) v = 0; }
    return v;
}
static bool use_igemm_convs() {      // GRAPHITE_CONV=igemm selects the im2col-chunk kernels for A/B runs
    static int v = -1;
    if (v < 0) { const char* e = getenv("GRAPHITE_CONV"); v = (e && strcmp(e, "igemm") == 0) ? 1 : 0; }
    return v == 1;
}

static int gtsrb_forward(gr_ctx* ctx, gr_model* m, const float* x, int N, float* logits, float* ws, hipStream_t st) {
    float* x4 = ws;
    float* A = x4 + (size_t)N * 4096;
    float* Bf = A + (size_t)N * 32768;
    if (x) {                                 // x == nullptr: the warp kernel already wrote relu(conv1_1(.)) into x4
        ProfScope ps_(ctx, "k_input_nhwc4<conv1_1>", st, 2.0 * N * 1024 * 9, (double)N * 1024 * 28);
        hipLaunchKernelGGL(k_input_nhwc4<true>, dim3((N * 1024 + 255) / 256), dim3(256), 0, st, x, m->c11, N, (float4*)x4);
    }
    GR_HIP(ctx, hipGetLastError());
    int rc;
    if (m->precision == GR_PREC_BF16) {
        const int mask = m->bf16_mask;
        // All layers in bf16 (the default): some of the activations between them are STORED as bf16 -- the values are
        // the ones the consumer would have rounded to anyway (same f32 number, same nearest-even rounding; max-pool
        // commutes with it), so the logits are bit-identical to the f32-activation pipeline (mask bit 7 keeps that one,
        // for the test).  A and Bf are reused as half-filled bf16 buffers.
        // links: bit i = the activation between layer i and layer i+1 (4: conv4_1 -> conv4_2, 5: conv4_2 -> fc5,
        // 6: fc5 -> fc6) is stored as bf16.  GRAPHITE_BF16_LINKS overrides the default (tuning).
        static int links_env = -1;
        if (links_env < 0) { const char* e = getenv("GRAPHITE_BF16_LINKS"); links_env = e ? (int)strtol(e, nullptr, 0) & 0x7f : BF16_LINKS_DEFAULT; }
        const int links = (mask & 0xff) == 0x7f ? links_env : 0;
        auto lk = [&](int i) { return ((links >> i) & 1) != 0; };
        auto conv = [&](int l, const float* in, int H, float* out) {
            return (mask >> l) & 1 ? run_gtsrb_halo_bf16(ctx, m->g[l], l, in, N, H, out, st, lk(l - 1), lk(l))
                                   : run_gtsrb_halo(ctx, m->g[l], l, in, N, H, out, st);
        };
        if (mask & 1) {
            Conv21BfParams q;
            q.in = x4; q.wgt = m->g[0].wbf; q.bias = m->g[0].bias; q.out = A; q.B = N;
            ProfScope ps_(ctx, "k_conv21_bf16<conv2_1>", st, 2.0 * N * 1024 * 32 * 75, 4.0 * N * 1024 * (3 + (lk(0) ? 16 : 32)));
            hipError_t e = lk(0) ? launch_conv21_bf16<16, true>(q, st) : launch_conv21_bf16<16, false>(q, st);
            if (e != hipSuccess) return gr_fail(ctx, GR_EHIP, "conv21_bf16 launch: %s", hipGetErrorString(e));
        } else if ((rc = run_gtsrb_halo(ctx, m->g[0], 0, x4, N, 32, A, st))) return rc;
        if ((rc = conv(1, A, 32, Bf))) return rc;
        if ((rc = conv(2, Bf, 16, A))) return rc;
        if ((rc = conv(3, A, 16, Bf))) return rc;
        if ((rc = conv(4, Bf, 8, A))) return rc;
        if ((rc = conv(5, A, 8, Bf))) return rc;
        if (mask & 0x40) {                                                                         // bit 6: fc5 / fc6
            if ((rc = run_fc_bf16(ctx, m->g[6], "fc5", Bf, N, A, st, lk(5), lk(6)))) return rc;
            if ((rc = run_fc_bf16(ctx, m->g[7], "fc6", A, N, Bf, st, lk(6), false))) return rc;     // fc7 (fp32) reads f32
        } else {
            if ((rc = run_conv(ctx, m->g[6], Bf, N, 1, 1, nullptr, nullptr, A, nullptr, true, false, st))) return rc;     // fc5
            if ((rc = run_conv(ctx, m->g[7], A, N, 1, 1, nullptr, nullptr, Bf, nullptr, true, false, st))) return rc;     // fc6
        }
        return run_conv(ctx, m->g[8], Bf, N, 1, 1, nullptr, nullptr, logits, nullptr, false, false, st);             // fc7
    }
    if (!use_igemm_convs()) {
        if ((rc = run_gtsrb_halo(ctx, m->g[0], 0, x4, N, 32, A, st))) return rc;
        if ((rc = run_gtsrb_halo(ctx, m->g[1], 1, A, N, 32, Bf, st))) return rc;
        if ((rc = run_gtsrb_halo(ctx, m->g[2], 2, Bf, N, 16, A, st))) return rc;
        if ((rc = run_gtsrb_halo(ctx, m->g[3], 3, A, N, 16, Bf, st))) return rc;
        if ((rc = run_gtsrb_halo(ctx, m->g[4], 4, Bf, N, 8, A, st))) return rc;
        if ((rc = run_gtsrb_halo(ctx, m->g[5], 5, A, N, 8, Bf, st))) return rc;
        if ((rc = run_conv(ctx, m->g[6], Bf, N, 1, 1, nullptr, nullptr, A, nullptr, true, false, st))) return rc;     // fc5
        if ((rc = run_conv(ctx, m->g[7], A, N, 1, 1, nullptr, nullptr, Bf, nullptr, true, false, st))) return rc;     // fc6
        return run_conv(ctx, m->g[8], Bf, N, 1, 1, nullptr, nullptr, logits, nullptr, false, false, st);             // fc7
    }
    if ((rc = run_conv(ctx, m->g[0], x4, N, 32, 32, nullptr, nullptr, A, nullptr, true, false, st))) return rc;   // conv2_1
    if ((rc = run_conv(ctx, m->g[1], A, N, 32, 32, nullptr, nullptr, Bf, nullptr, true, true, st))) return rc;    // conv2_2+pool
    if ((rc = run_conv(ctx, m->g[2], Bf, N, 16, 16, nullptr, nullptr, A, nullptr, true, false, st))) return rc;   // conv3_1
    if ((rc = run_conv(ctx, m->g[3], A, N, 16, 16, nullptr, nullptr, Bf, nullptr, true, true, st))) return rc;    // conv3_2+pool
    if ((rc = run_conv(ctx, m->g[4], Bf, N, 8, 8, nullptr, nullptr, A, nullptr, true, false, st))) return rc;     // conv4_1
    if ((rc = run_conv(ctx, m->g[5], A, N, 8, 8, nullptr, nullptr, Bf, nullptr, true, true, st))) return rc;      // conv4_2+pool
    if ((rc = run_conv(ctx, m->g[6], Bf, N, 1, 1, nullptr, nullptr, A, nullptr, true, false, st))) return rc;     // fc5
    if ((rc = run_conv(ctx, m->g[7], A, N, 1, 1, nullptr, nullptr, Bf, nullptr, true, false, st))) return rc;     // fc6
    return run_conv(ctx, m->g[8], Bf, N, 1, 1, nullptr, nullptr, logits, nullptr, false, false, st);             // fc7
}

static int wrn_forward(gr_ctx* ctx, gr_model* m, const float* x, int N, float* logits, float* ws, hipStream_t st) {
    size_t big = (size_t)1024 * m->ch[1];
    if ((size_t)256 * m->ch[2] > big) big = (size_t)256 * m->ch[2];
    float* x4 = ws;
    float* xr = x4 + (size_t)N * 4096;       // raw block input (residual)
    float* o = xr + (size_t)N * big;         // relu(bn1(x))
    float* u = o + (size_t)N * big;          // relu(bn2(conv1))
    float* s = u + (size_t)N * big;          // shortcut conv
    float* y = s + (size_t)N * big;          // next raw
    if (x)
        hipLaunchKernelGGL(k_input_nhwc4<false>, dim3((N * 1024 + 255) / 256), dim3(256), 0, st, x, (const float*)nullptr, N,
                           (float4*)x4);
    GR_HIP(ctx, hipGetLastError());
    int rc;
    const size_t nb = m->blocks.size();
    // conv1 -> raw x (block1.0 needs it only if equalInOut) and o = relu(bn1_0(x))
    if ((rc = run_conv(ctx, m->wconv1, x4, N, 32, 32, nullptr, &m->blocks[0].bn1, m->blocks[0].has_sc ? nullptr : xr, o,
                       false, false, st))) return rc;
    int H = 32;
    bool o_bf = false;                       // does `o` currently hold bf16?
    for (size_t bi = 0; bi < nb; bi++) {
        WrnBlock& b = m->blocks[bi];
        int Ho = H / b.stride;
        // bf16 mode: an activation is STORED as bf16 when the bf16 kernel produces it and only bf16 kernels read it (u:
        // conv1 -> conv2; o: conv2 -> the next block's conv1, unless that block also feeds o to its f32 1x1 shortcut
        // or it is the last one, which the average pool reads).  Same values the consumer would round to: the logits
        // do not change by a bit (model mask bit 7 keeps f32, for the test).
        const bool abf = m->precision == GR_PREC_BF16 && !(m->bf16_mask & 0x80);
        const bool u_bf = abf && cc_bf16_ok(b.conv1, H, m->precision) && cc_bf16_ok(b.conv2, Ho, m->precision);
        if ((rc = run_conv(ctx, b.conv1, o, N, H, H, nullptr, &b.bn2, nullptr, u, false, false, st, nullptr, m->precision, o_bf, u_bf))) return rc;
        const float* res = xr;
        if (b.has_sc) {
            if ((rc = run_conv(ctx, b.sc, o, N, H, H, nullptr, nullptr, s, nullptr, false, false, st))) return rc;
            res = s;
        }
        bool last = bi + 1 == nb;
        const BnDev* nbn = last ? &m->bnf : &m->blocks[bi + 1].bn1;
        bool need_raw = !last && !m->blocks[bi + 1].has_sc;
        // o is free once conv1 and the shortcut have consumed it: the activated output goes there
        const bool next_o_bf = abf && !last && cc_bf16_ok(b.conv2, Ho, m->precision) && !m->blocks[bi + 1].has_sc &&
                               m->blocks[bi + 1].stride == 1 && cc_bf16_ok(m->blocks[bi + 1].conv1, Ho, m->precision);
        if ((rc = run_conv(ctx, b.conv2, u, N, Ho, Ho, res, nbn, need_raw ? y : nullptr, o, false, false, st, nullptr, m->precision, u_bf, next_o_bf))) return rc;
        o_bf = next_o_bf;
        if (need_raw) { float* tmp = xr; xr = y; y = tmp; }
        H = Ho;
    }
    int C = m->ch[3];
    hipLaunchKernelGGL(k_avgpool, dim3((N * C + 255) / 256), dim3(256), 0, st, o, N, H * H, C, u);
    GR_HIP(ctx, hipGetLastError());
    return run_conv(ctx, m->wfc, u, N, 1, 1, nullptr, nullptr, logits, nullptr, false, false, st);
}

// ---------------------------------------------------------------------------------------------
// input gradient of GTSRBNet (the classifier half of the whitebox EOT loop, whitebox_attack.py:155-158)
// ---------------------------------------------------------------------------------------------
// 2x2 max-pool of an NHWC tensor [N,H,W,C] -> [N,H/2,W/2,C]
__global__ void k_maxpool2_fwd(const float* __restrict__ in, int N, int H, int W, int C, float* __restrict__ out) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int Hp = H >> 1, Wp = W >> 1;
    if (idx >= (size_t)N * Hp * Wp * C) return;
    const int c = (int)(idx % C);
    size_t r = idx / C;
    const int px = (int)(r % Wp); r /= Wp;
    const int py = (int)(r % Hp);
    const int n = (int)(r / Hp);
    const float* s = in + (((size_t)n * H + 2 * py) * W + 2 * px) * C + c;
    const float v0 = s[0], v1 = s[C], v2 = s[(size_t)W * C], v3 = s[(size_t)W * C + C];
    float mx = v0 > v1 ? v0 : v1;
    mx = mx > v2 ? mx : v2;
    out[idx] = mx > v3 ? mx : v3;
}
// backward of relu -> max-pool: d_pre[pixel] = d_pool[window] for the FIRST pixel of the window (row-major scan, as
// torch's max_pool2d picks) whose activation equals the window maximum, and only where that activation is > 0 (ReLU)
__global__ void k_maxpool2_relu_bwd(const float* __restrict__ act, const float* __restrict__ gpool, int N, int H, int W, int C,
                                    float* __restrict__ gpre) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int Hp = H >> 1, Wp = W >> 1;
    if (idx >= (size_t)N * Hp * Wp * C) return;
    const int c = (int)(idx % C);
    size_t r = idx / C;
    const int px = (int)(r % Wp); r /= Wp;
    const int py = (int)(r % Hp);
    const int n = (int)(r / Hp);
    const size_t base = (((size_t)n * H + 2 * py) * W + 2 * px) * C + c;
    const size_t off[4] = {0, (size_t)C, (size_t)W * C, (size_t)W * C + C};
    float v[4];
    for (int k = 0; k < 4; k++) v[k] = act[base + off[k]];
    int best = 0;
    for (int k = 1; k < 4; k++)
        if (v[k] > v[best]) best = k;
    const float g = v[best] > 0.f ? gpool[idx] : 0.f;
    for (int k = 0; k < 4; k++) gpre[base + off[k]] = k == best ? g : 0.f;
}
// d CE(logits, target) / d logits = softmax(logits) - onehot(target), written into a 64-wide row (zero padded)
__global__ void k_ce_grad(const float* __restrict__ logits, int N, int ncls, int target, float* __restrict__ g64) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float* l = logits + (size_t)n * ncls;
    float mx = l[0];
    for (int c = 1; c < ncls; c++) mx = l[c] > mx ? l[c] : mx;
    float sum = 0.f;
    for (int c = 0; c < ncls; c++) sum += expf(l[c] - mx);
    float* g = g64 + (size_t)n * 64;
    for (int c = 0; c < 64; c++) g[c] = c < ncls ? expf(l[c] - mx) / sum - (c == target ? 1.f : 0.f) : 0.f;
}
// backward of k_input_nhwc4<true>: ReLU gate on the stored conv1_1 output, 3x3 transpose, NHWC3 -> CHW
__global__ void k_input_grad(const float* __restrict__ g3, const float4* __restrict__ x4, const float* __restrict__ c11, int N,
                             float* __restrict__ gx) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * 1024) return;
    const int n = idx >> 10, p = idx & 1023;
    const float4 a = x4[idx];
    const float* g = g3 + (size_t)idx * 3;
    const float r0 = a.x > 0.f ? g[0] : 0.f, r1 = a.y > 0.f ? g[1] : 0.f, r2 = a.z > 0.f ? g[2] : 0.f;
    float* o = gx + (size_t)n * 3072 + p;
#pragma unroll
    for (int i = 0; i < 3; i++) o[i * 1024] = r0 * c11[0 * 3 + i] + r1 * c11[1 * 3 + i] + r2 * c11[2 * 3 + i];
}

static size_t grad_floats_per_image() {
    //     x4     a21     a22     p2     a31     a32     p3     a41    a42    p4     f5     f6    lg   g64   gA      gB
    return 4096 + 32768 + 32768 + 8192 + 16384 + 16384 + 4096 + 8192 + 8192 + 2048 + 1024 + 1024 + 64 + 64 + 32768 + 32768;
}
static size_t wrn_grad_floats_per_image(const gr_model* m);
static int wrn_input_grad(gr_ctx* ctx, gr_model* m, const float* x, int N, int target, float* logits_out, float* gx,
                          float* w, hipStream_t st);
size_t net_grad_workspace_bytes(const gr_model* m, int N) {
    const size_t per = m->kind == 2 ? wrn_grad_floats_per_image(m) : grad_floats_per_image();
    return gr_align(per * sizeof(float) * (size_t)N);
}

// logits (optional) and d CE(logits_n, target) / d x_n for N images; GTSRBNet only.
int net_input_grad(gr_ctx* ctx, gr_model* m, const float* x, int N, int target, float* logits_out, float* gx, char* wsb,
                   size_t ws_bytes, hipStream_t st) {
    if (N <= 0) return GR_OK;
    if (target < 0 || target >= m->ncls) return gr_fail(ctx, GR_EINVAL, "target %d out of range", target);
    if (ws_bytes < net_grad_workspace_bytes(m, N)) return gr_fail(ctx, GR_ESTATE, "net_input_grad: workspace too small");
    if (m->kind == 2) return wrn_input_grad(ctx, m, x, N, target, logits_out, gx, (float*)wsb, st);
    if (m->kind != 1) return gr_fail(ctx, GR_EINVAL, "unknown model kind");
    float* w = (float*)wsb;
    auto take = [&](size_t per) { float* p = w; w += per * (size_t)N; return p; };
    float *x4 = take(4096), *a21 = take(32768), *a22 = take(32768), *p2 = take(8192), *a31 = take(16384), *a32 = take(16384);
    float *p3 = take(4096), *a41 = take(8192), *a42 = take(8192), *p4 = take(2048), *f5 = take(1024), *f6 = take(1024);
    float *lg = take(64), *g64 = take(64), *gA = take(32768), *gB = take(32768);
    auto blocks = [](size_t n) { return dim3((unsigned)((n + 255) / 256)); };
    int rc;
    // ---- forward, keeping every activation (pool layers un-fused so that the arg-max is recoverable) ----
    hipLaunchKernelGGL(k_input_nhwc4<true>, blocks((size_t)N * 1024), dim3(256), 0, st, x, m->c11, N, (float4*)x4);
    if ((rc = run_conv(ctx, m->g[0], x4, N, 32, 32, nullptr, nullptr, a21, nullptr, true, false, st))) return rc;
    if ((rc = run_conv(ctx, m->g[1], a21, N, 32, 32, nullptr, nullptr, a22, nullptr, true, false, st))) return rc;
    hipLaunchKernelGGL(k_maxpool2_fwd, blocks((size_t)N * 8192), dim3(256), 0, st, a22, N, 32, 32, 32, p2);
    if ((rc = run_conv(ctx, m->g[2], p2, N, 16, 16, nullptr, nullptr, a31, nullptr, true, false, st))) return rc;
    if ((rc = run_conv(ctx, m->g[3], a31, N, 16, 16, nullptr, nullptr, a32, nullptr, true, false, st))) return rc;
    hipLaunchKernelGGL(k_maxpool2_fwd, blocks((size_t)N * 4096), dim3(256), 0, st, a32, N, 16, 16, 64, p3);
    if ((rc = run_conv(ctx, m->g[4], p3, N, 8, 8, nullptr, nullptr, a41, nullptr, true, false, st))) return rc;
    if ((rc = run_conv(ctx, m->g[5], a41, N, 8, 8, nullptr, nullptr, a42, nullptr, true, false, st))) return rc;
    hipLaunchKernelGGL(k_maxpool2_fwd, blocks((size_t)N * 2048), dim3(256), 0, st, a42, N, 8, 8, 128, p4);
    if ((rc = run_conv(ctx, m->g[6], p4, N, 1, 1, nullptr, nullptr, f5, nullptr, true, false, st))) return rc;
    if ((rc = run_conv(ctx, m->g[7], f5, N, 1, 1, nullptr, nullptr, f6, nullptr, true, false, st))) return rc;
    float* lgp = logits_out ? logits_out : lg;
    if ((rc = run_conv(ctx, m->g[8], f6, N, 1, 1, nullptr, nullptr, lgp, nullptr, false, false, st))) return rc;
    // ---- backward: every transposed layer gates its output with the ReLU of the activation it produces the gradient of ----
    hipLaunchKernelGGL(k_ce_grad, blocks((size_t)N), dim3(256), 0, st, lgp, N, m->ncls, target, g64);
    if ((rc = run_conv(ctx, m->gb[8], g64, N, 1, 1, nullptr, nullptr, gA, nullptr, false, false, st, f6))) return rc;   // d f6
    if ((rc = run_conv(ctx, m->gb[7], gA, N, 1, 1, nullptr, nullptr, gB, nullptr, false, false, st, f5))) return rc;    // d f5
    if ((rc = run_conv(ctx, m->gb[6], gB, N, 1, 1, nullptr, nullptr, gA, nullptr, false, false, st))) return rc;        // d p4
    hipLaunchKernelGGL(k_maxpool2_relu_bwd, blocks((size_t)N * 2048), dim3(256), 0, st, a42, gA, N, 8, 8, 128, gB);      // d a42
    if ((rc = run_conv(ctx, m->gb[5], gB, N, 8, 8, nullptr, nullptr, gA, nullptr, false, false, st, a41))) return rc;    // d a41
    if ((rc = run_conv(ctx, m->gb[4], gA, N, 8, 8, nullptr, nullptr, gB, nullptr, false, false, st))) return rc;         // d p3
    hipLaunchKernelGGL(k_maxpool2_relu_bwd, blocks((size_t)N * 4096), dim3(256), 0, st, a32, gB, N, 16, 16, 64, gA);     // d a32
    if ((rc = run_conv(ctx, m->gb[3], gA, N, 16, 16, nullptr, nullptr, gB, nullptr, false, false, st, a31))) return rc;  // d a31
    if ((rc = run_conv(ctx, m->gb[2], gB, N, 16, 16, nullptr, nullptr, gA, nullptr, false, false, st))) return rc;       // d p2
    hipLaunchKernelGGL(k_maxpool2_relu_bwd, blocks((size_t)N * 8192), dim3(256), 0, st, a22, gA, N, 32, 32, 32, gB);     // d a22
    if ((rc = run_conv(ctx, m->gb[1], gB, N, 32, 32, nullptr, nullptr, gA, nullptr, false, false, st, a21))) return rc;  // d a21
    if ((rc = run_conv(ctx, m->gb[0], gA, N, 32, 32, nullptr, nullptr, gB, nullptr, false, false, st))) return rc;       // d x4 [M,3]
    hipLaunchKernelGGL(k_input_grad, blocks((size_t)N * 1024), dim3(256), 0, st, gB, (const float4*)x4, m->c11, N, gx);
    GR_HIP(ctx, hipGetLastError());
    return GR_OK;
}

// ---------------------------------------------------------------------------------------------
// input gradient of WideResNet (whitebox_attack.py:155-158 with the CIFAR classifier, cifar/wideresnet.py:25-34,86-94)
// ---------------------------------------------------------------------------------------------
// backward of relu(bn(.)) in eval mode: g_pre = g_post * scale[c] where the activation is positive; optional + add[idx]
__global__ void k_bn_relu_bwd(const float* __restrict__ g, const float* __restrict__ act, const float* __restrict__ scale,
                              const float* __restrict__ add, size_t n, int C, float* __restrict__ out) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int c = (int)(idx % C);
    float v = act[idx] > 0.f ? g[idx] * scale[c] : 0.f;
    if (add) v = v + add[idx];
    out[idx] = v;
}
// zero-dilation of a gradient for the transposed pass of a stride-2 layer: out[n, 2y, 2x, c] = g[n, y, x, c], 0 elsewhere
__global__ void k_dilate2(const float* __restrict__ g, int N, int Ho, int Wo, int C, float* __restrict__ out) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int H = 2 * Ho, W = 2 * Wo;
    if (idx >= (size_t)N * H * W * C) return;
    const int c = (int)(idx % C);
    size_t r = idx / C;
    const int x = (int)(r % W); r /= W;
    const int y = (int)(r % H);
    const int n = (int)(r / H);
    out[idx] = ((x | y) & 1) ? 0.f : g[(((size_t)n * Ho + (y >> 1)) * Wo + (x >> 1)) * C + c];
}
// backward of avg_pool2d(P pixels) after relu(bn(.)): g_y[n, p, c] = g_pooled[n, c] / P * scale[c] where act > 0
__global__ void k_pool_bn_relu_bwd(const float* __restrict__ gp, const float* __restrict__ act, const float* __restrict__ scale,
                                   int N, int P, int C, float* __restrict__ out) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)N * P * C) return;
    const int c = (int)(idx % C);
    const int n = (int)(idx / ((size_t)P * C));
    out[idx] = act[idx] > 0.f ? (gp[(size_t)n * C + c] * (1.0f / (float)P)) * scale[c] : 0.f;
}
// [N, 1024, 3 of CP channels] NHWC -> [N, 3, 1024] CHW
__global__ void k_nhwc_to_chw3(const float* __restrict__ g, int N, int CP, float* __restrict__ gx) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * 1024) return;
    const int n = idx >> 10, p = idx & 1023;
    for (int c = 0; c < 3; c++) gx[(size_t)n * 3072 + c * 1024 + p] = g[(size_t)idx * CP + c];
}

static size_t wrn_big(const gr_model* m) {
    size_t big = (size_t)1024 * m->ch[1];
    if ((size_t)256 * m->ch[2] > big) big = (size_t)256 * m->ch[2];
    if ((size_t)64 * m->ch[3] > big) big = (size_t)64 * m->ch[3];
    return big;
}
// gradient buffers also hold the zero-dilated gradient of a strided block: [Hi, Hi, cout] with Hi = 2 Ho
static size_t wrn_gbig(const gr_model* m) {
    size_t g = wrn_big(m);
    int H = 32;
    for (const WrnBlock& b : m->blocks) {
        if (b.stride == 2 && (size_t)H * H * b.cout > g) g = (size_t)H * H * b.cout;
        H /= b.stride;
    }
    return g;
}
static size_t wrn_grad_floats_per_image(const gr_model* m) {
    size_t keep = 0;
    int H = 32;
    for (const WrnBlock& b : m->blocks) {
        const int Ho = H / b.stride;
        keep += (size_t)H * H * b.cin + (size_t)Ho * Ho * b.cout;      // o = relu(bn1(x)), u = relu(bn2(conv1(o)))
        H = Ho;
    }
    keep += (size_t)H * H * m->ch[3];                                   // relu(bn(y_last))
    //     x4    raw x ping-pong + shortcut   stored acts  pooled      logits g64   gradient buffers (4) + dilated
    return 4096 + 3 * wrn_big(m) + keep + m->ch[3] + 64 + 64 + 5 * wrn_gbig(m);
}

static int wrn_input_grad(gr_ctx* ctx, gr_model* m, const float* x, int N, int target, float* logits_out, float* gx,
                          float* w, hipStream_t st) {
    const size_t big = wrn_big(m);
    auto take = [&](size_t per) { float* p = w; w += per * (size_t)N; return p; };
    auto blocks = [](size_t n) { return dim3((unsigned)((n + 255) / 256)); };
    float* x4 = take(4096);
    float *xr = take(big), *y = take(big), *s = take(big);
    const size_t nb = m->blocks.size();
    std::vector<float*> O(nb), U(nb);
    {
        int H = 32;
        for (size_t bi = 0; bi < nb; bi++) {
            const WrnBlock& b = m->blocks[bi];
            const int Ho = H / b.stride;
            O[bi] = take((size_t)H * H * b.cin);
            U[bi] = take((size_t)Ho * Ho * b.cout);
            H = Ho;
        }
    }
    float* OF = take((size_t)64 * m->ch[3]);
    float *pooled = take(m->ch[3]), *lg = take(64), *g64 = take(64);
    const size_t gbig = wrn_gbig(m);
    float *gY = take(gbig), *gA = take(gbig), *gB = take(gbig), *gS = take(gbig), *dil = take(gbig);
    int rc;
    // ---- forward, every post-activation kept (fp32 kernels, whatever gr_model_set_precision says) ----
    hipLaunchKernelGGL(k_input_nhwc4<false>, blocks((size_t)N * 1024), dim3(256), 0, st, x, (const float*)nullptr, N, (float4*)x4);
    if ((rc = run_conv(ctx, m->wconv1, x4, N, 32, 32, nullptr, &m->blocks[0].bn1, m->blocks[0].has_sc ? nullptr : xr, O[0],
                       false, false, st))) return rc;
    int H = 32;
    for (size_t bi = 0; bi < nb; bi++) {
        WrnBlock& b = m->blocks[bi];
        const int Ho = H / b.stride;
        if ((rc = run_conv(ctx, b.conv1, O[bi], N, H, H, nullptr, &b.bn2, nullptr, U[bi], false, false, st))) return rc;
        const float* res = xr;
        if (b.has_sc) {
            if ((rc = run_conv(ctx, b.sc, O[bi], N, H, H, nullptr, nullptr, s, nullptr, false, false, st))) return rc;
            res = s;
        }
        const bool last = bi + 1 == nb;
        const BnDev* nbn = last ? &m->bnf : &m->blocks[bi + 1].bn1;
        const bool need_raw = !last && !m->blocks[bi + 1].has_sc;
        if ((rc = run_conv(ctx, b.conv2, U[bi], N, Ho, Ho, res, nbn, need_raw ? y : nullptr, last ? OF : O[bi + 1], false, false, st)))
            return rc;
        if (need_raw) { float* tmp = xr; xr = y; y = tmp; }
        H = Ho;
    }
    const int C3 = m->ch[3], P = H * H;
    hipLaunchKernelGGL(k_avgpool, blocks((size_t)N * C3), dim3(256), 0, st, OF, N, P, C3, pooled);
    float* lgp = logits_out ? logits_out : lg;
    if ((rc = run_conv(ctx, m->wfc, pooled, N, 1, 1, nullptr, nullptr, lgp, nullptr, false, false, st))) return rc;
    // ---- backward ----
    hipLaunchKernelGGL(k_ce_grad, blocks((size_t)N), dim3(256), 0, st, lgp, N, m->ncls, target, g64);
    if ((rc = run_conv(ctx, m->wfcT, g64, N, 1, 1, nullptr, nullptr, gA, nullptr, false, false, st))) return rc;       // d pooled
    hipLaunchKernelGGL(k_pool_bn_relu_bwd, blocks((size_t)N * P * C3), dim3(256), 0, st, gA, OF, m->bnf.scale, N, P, C3, gY);
    for (size_t bi = nb; bi-- > 0;) {
        WrnBlock& b = m->blocks[bi];
        const int Ho = H, Hi = H * b.stride;             // block output / input size
        const size_t mo = (size_t)N * Ho * Ho, mi = (size_t)N * Hi * Hi;
        // d u = conv2^T(d y);  d conv1 = d u * bn2.scale where u > 0
        if ((rc = run_conv(ctx, b.conv2T, gY, N, Ho, Ho, nullptr, nullptr, gA, nullptr, false, false, st))) return rc;
        hipLaunchKernelGGL(k_bn_relu_bwd, blocks(mo * b.cout), dim3(256), 0, st, gA, U[bi], b.bn2.scale, (const float*)nullptr,
                           mo * b.cout, b.cout, gB);
        // d o = conv1^T(d conv1) [+ shortcut^T(d y)], strided layers on the zero-dilated gradients
        const float* g1 = gB;
        const float* gy_in = gY;
        if (b.stride == 2) {
            hipLaunchKernelGGL(k_dilate2, blocks(mi * b.cout), dim3(256), 0, st, gB, N, Ho, Ho, b.cout, dil);
            g1 = dil;
        }
        const float* res = nullptr;
        if (b.has_sc) {
            if (b.stride == 2) {
                hipLaunchKernelGGL(k_dilate2, blocks(mi * b.cout), dim3(256), 0, st, gY, N, Ho, Ho, b.cout, gA);
                gy_in = gA;
            }
            if ((rc = run_conv(ctx, b.scT, gy_in, N, Hi, Hi, nullptr, nullptr, gS, nullptr, false, false, st))) return rc;
            res = gS;
        }
        float* gO = (g1 == dil) ? gB : gA;               // a buffer neither operand lives in
        if (b.has_sc && b.stride == 2) gO = gB;          // gA holds the dilated d y until scT has run (it has)
        if ((rc = run_conv(ctx, b.conv1T, g1, N, Hi, Hi, res, nullptr, gO, nullptr, false, false, st))) return rc;
        // d x = d o * bn1.scale where o > 0 (+ d y through the identity shortcut)
        float* gX = (gO == gA) ? gB : gA;
        hipLaunchKernelGGL(k_bn_relu_bwd, blocks(mi * b.cin), dim3(256), 0, st, gO, O[bi], b.bn1.scale,
                           b.has_sc ? (const float*)nullptr : (const float*)gY, mi * b.cin, b.cin, gX);
        float* old = gY; gY = gX;
        if (gX == gA) gA = old; else gB = old;           // keep the three buffers distinct
        H = Hi;
    }
    // d input = conv1^T(d x): [N,1024,3] (NP channels per pixel) -> CHW
    if ((rc = run_conv(ctx, m->wconv1T, gY, N, 32, 32, nullptr, nullptr, gA, nullptr, false, false, st))) return rc;
    hipLaunchKernelGGL(k_nhwc_to_chw3, blocks((size_t)N * 1024), dim3(256), 0, st, gA, N, 3, gx);
    GR_HIP(ctx, hipGetLastError());
    return GR_OK;
}

// local logits [N,h,h,ncls] of a PatchGuard model (patchguard_predict.py:42-44): pre-processing + conv1 in one kernel, then
// the folded conv+BN layers on the fp32 MFMA implicit-GEMM kernel (1x1 convs are plain GEMMs over N*H*W rows)
int net_pg_local(gr_ctx* ctx, gr_model* m, const float* x, int N, float* local, char* wsb, size_t ws_bytes, hipStream_t st) {
    if (N <= 0) return GR_OK;
    if (m->kind != 3) return gr_fail(ctx, GR_EINVAL, "not a PatchGuard model");
    if (ws_bytes < net_workspace_bytes(m, N)) return gr_fail(ctx, GR_ESTATE, "net_pg_local: workspace too small");
    float* ws = (float*)wsb;
    float* x4 = ws;
    float* X = x4 + (size_t)N * 4096;
    float* T1 = X + (size_t)N * m->pg_big;
    float* T2 = T1 + (size_t)N * m->pg_big;
    float* R = T2 + (size_t)N * m->pg_big;
    int rc;
    const int D = m->pg_D;
    // GR_PREC_BF16 (throughput mode, not the parity path): the stride-1 1x1 convs -- 74 % of bagnet17's flops -- run as bf16
    // GEMMs with f32 accumulation; activations stay f32 in HBM (rounded while a stage is loaded), the 3x3 convs, the strided
    // shortcuts and the fc stay on the fp32 kernel.
    auto conv = [&](const Layer& L, const float* in, int H, const float* res, float* out, bool relu, int out_hw) -> int {
        if (m->precision == GR_PREC_BF16 && L.wbf && L.KH == 1 && L.stride == 1 && out_hw == 0) {
            GemmBfParams q;
            q.a = in; q.w = L.wbf; q.bias = L.bias; q.c = out; q.res = res;
            q.M = N * H * H; q.N = L.Cout; q.K = L.CinT; q.relu = relu ? 1 : 0;
            char pname[64];
            snprintf(pname, sizeof(pname), "k_gemm_bf16<bagnet N%d>", L.Cout >= 128 ? 128 : 64);
            ProfScope ps_(ctx, pname, st, 2.0 * q.M * q.N * q.K, 4.0 * q.M * ((double)q.K + q.N * (res ? 2 : 1)) + 2.0 * q.K * q.N);
            hipError_t e;
            if (L.Cout % 128 == 0) e = q.K % 64 == 0 && q.K >= 128 ? launch_gemm_bf16<2, 2, 2, 2, 64>(q, st) : launch_gemm_bf16<2, 2, 2, 2, 32>(q, st);
            else e = q.K % 64 == 0 && q.K >= 128 ? launch_gemm_bf16<2, 2, 2, 1, 64>(q, st) : launch_gemm_bf16<2, 2, 2, 1, 32>(q, st);
            if (e != hipSuccess) return gr_fail(ctx, GR_EHIP, "gemm_bf16 launch (bagnet %dx%d): %s", q.K, q.N, hipGetErrorString(e));
            return GR_OK;
        }
        return run_conv(ctx, L, in, N, H, H, res, nullptr, out, nullptr, relu, false, st, nullptr, GR_PREC_F32, false, false, out_hw);
    };
    if ((rc = pg_preprocess(ctx, x ? x : x4, N, 32, D, x == nullptr, m->bag_conv1, T1, st))) return rc;
    if ((rc = run_conv(ctx, m->bag_conv2, T1, N, D, D, nullptr, nullptr, X, nullptr, true, false, st))) return rc;
    int H = D - 2;
    for (BagBlock& b : m->bag) {
        const int Ho = (H - b.k) / b.stride + 1;
        if ((rc = conv(b.c1, X, H, nullptr, T1, true, 0))) return rc;
        if ((rc = conv(b.c2, T1, H, nullptr, T2, true, 0))) return rc;
        const float* res = X;
        if (b.has_ds) {             // 1x1 stride-s conv + BN; only the Ho x Ho corner the crop keeps
            if ((rc = conv(b.ds, X, H, nullptr, R, false, Ho == (H - 1) / b.stride + 1 ? 0 : Ho))) return rc;
            res = R;
        } else if (Ho != H) return gr_fail(ctx, GR_ESTATE, "bagnet: identity residual of a different size");
        if ((rc = conv(b.c3, T2, Ho, res, T1, true, 0))) return rc;
        float* tmp = X; X = T1; T1 = tmp;
        H = Ho;
    }
    return run_conv(ctx, m->bag_fc, X, N, H, H, nullptr, nullptr, local, nullptr, false, false, st);
}

static int pg_forward(gr_ctx* ctx, gr_model* m, const float* x, int N, float* logits, char* ws, size_t ws_bytes, hipStream_t st) {
    float* local = (float*)ws + (size_t)N * (4096 + 4 * m->pg_big);
    int rc = net_pg_local(ctx, m, x, N, local, ws, ws_bytes, st);
    if (rc) return rc;
    if (m->pg_defense) return pg_masking(ctx, local, N, m->pg_h, m->pg_h, m->ncls, m->pg_window, m->pg_window, m->pg_thres, logits, nullptr, st);
    return pg_mean(ctx, local, N, m->pg_h, m->pg_h, m->ncls, logits, st);
}

int net_input_mode(const gr_model* m, const float** c11) {
    if (m->kind == 1) { *c11 = m->c11; return GR_OUT_NHWC4_C11; }
    *c11 = nullptr;
    return GR_OUT_NHWC4;
}

int net_forward(gr_ctx* ctx, gr_model* m, const float* x, int N, float* logits, char* ws, size_t ws_bytes, hipStream_t st) {
    if (N <= 0) return GR_OK;
    if (ws_bytes < net_workspace_bytes(m, N)) return gr_fail(ctx, GR_ESTATE, "net_forward: workspace too small");
    if (m->kind == 1) return gtsrb_forward(ctx, m, x, N, logits, (float*)ws, st);
    if (m->kind == 2) return wrn_forward(ctx, m, x, N, logits, (float*)ws, st);
    if (m->kind == 3) return pg_forward(ctx, m, x, N, logits, ws, ws_bytes, st);
    return gr_fail(ctx, GR_ESTATE, "unknown model kind");
}

int net_hits(gr_ctx* ctx, const float* logits, int N, int ncls, int64_t n_begin, int T, int target, int32_t* hits,
             int64_t* preds, hipStream_t st, int tie_last) {
    if (N <= 0) return GR_OK;
    ProfScope ps_(ctx, "k_argmax_hits", st, 0.0, (double)N * ncls * 4);
    hipLaunchKernelGGL(k_argmax_hits, dim3((N + 255) / 256), dim3(256), 0, st, logits, N, ncls, (long long)n_begin, T,
                       target, (int*)hits, (long long*)preds, tie_last);
    GR_HIP(ctx, hipGetLastError());
    return GR_OK;
}
int net_argmax(gr_ctx* ctx, const float* logits, int N, int ncls, int64_t* preds, hipStream_t st, int tie_last) {
    return net_hits(ctx, logits, N, ncls, 0, 1, -1, nullptr, preds, st, tie_last);
}
int net_tie_last(const gr_model* m) { return (m && m->kind == 3 && m->pg_defense) ? 1 : 0; }
