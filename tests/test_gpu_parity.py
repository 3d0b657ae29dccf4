"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bars (north star): warped pixels, labels and hit counts bit-exact vs the oracle (the kernels use the
oracle's operation order; fp32 MFMA is a k-ordered fmaf chain); logits within 1e-4 relative of the
reference's own PyTorch forward (golden nets.npz).
"""
import os

import numpy as np
import pytest
import torch

from cases import GOLD, gtsrb_case, gtsrb_candidates, cifar_case
from graphite_amd import _hip, synth, weights, transforms as T, utils, models, engine

pytestmark = pytest.mark.gpu

REL_TOL = 1e-4          # logits, relative to the largest |logit| (north star)


@pytest.fixture(scope="module")
def gtsrb_model():
    m = models.GTSRBNet()
    m.load_state_dict(synth.gtsrb_state_dict())
    return m


@pytest.fixture(scope="module")
def gtsrb_tensors():
    return weights.gtsrb_tensor_list(synth.gtsrb_state_dict())


def test_native_library_is_loaded():
    _hip.ctx()
    maps = open("/proc/self/maps").read()
    assert "libgraphite_hip.so" in maps


# ------------------------------------------------------------------------------ pixel operators
def test_resize_bit_exact(oracle):
    x = synth.hash_uniform("rs", (3, 32, 32))
    for (dh, dw) in ((244, 244), (32, 32), (100, 37)):
        assert np.array_equal(_hip.resize_linear(x, dh, dw), oracle.resize_linear(x, dh, dw))
    big = synth.hash_uniform("rs-big", (3, 244, 244))
    for s in (32, 47, 112):
        assert np.array_equal(_hip.resize_linear(big, s, s), oracle.resize_linear(big, s, s))


def test_add_noise_bit_exact(oracle):
    case = gtsrb_case()
    masks, thetas = gtsrb_candidates(case, 3)
    for c in range(3):
        for lbd, clip in ((1.0, True), (0.5, False), (0.0, True)):
            got = _hip.add_noise(case["vic"], masks[c], thetas[c], lbd, clip)
            ref = oracle.add_noise(case["vic"], masks[c], thetas[c], lbd, clip)
            for g, r in zip(got, ref):
                assert np.array_equal(g, r)
    t = T.add_noise(torch.from_numpy(case["vic"]), torch.from_numpy(masks[0]), 1.0, torch.from_numpy(thetas[0]), True)
    assert t[2].dtype == torch.float64 and t[0].dtype == torch.float32


def test_warp_perspective_bit_exact(oracle):
    case = gtsrb_case()
    for t in (0, 3, 17, 55):
        xf = case["tab"][t]
        _, ref = oracle.warp_one(case["vic"], xf, want_warped=True)
        got = _hip.warp_perspective_hwc(np.transpose(case["vic"], (1, 2, 0)), xf["M"].reshape(3, 3), int(xf["crop_x"]),
                                        int(xf["crop_y"]))
        assert np.array_equal(np.transpose(got, (2, 0, 1)), ref)


def test_warp_batch_gtsrb_bit_exact(oracle):
    case = gtsrb_case()
    masks, thetas = gtsrb_candidates(case, 4)
    got = _hip.warp_batch(case["vic"], masks, thetas, case["tab"], 1.0, True)
    ref = oracle.warp_batch(case["vic"], masks, thetas, case["tab"])
    assert got.shape == ref.shape == (4, 100, 3, 32, 32)
    diff = np.abs(got - ref)
    assert diff.max() <= 1.0 / 255 + 1e-6                   # north-star bar
    assert np.array_equal(got, ref), f"{(diff > 0).sum()} pixels differ, max {diff.max()}"


def test_warp_batch_blur_kernels_and_baseline(oracle):
    case = gtsrb_case(baseline=True, num_xforms=40)
    masks, thetas = gtsrb_candidates(case, 2)
    got = _hip.warp_batch(case["vic"], masks, thetas, case["tab"], 1.0, True)
    assert np.array_equal(got, oracle.warp_batch(case["vic"], masks, thetas, case["tab"]))
    # single-kernel tables (one composite slot each)
    full = gtsrb_case()
    for k in (0, 3, 5, 7):
        tab = full["tab"][full["tab"]["blur"] == k][:6]
        got = _hip.warp_batch(full["vic"], masks[:1], thetas[:1], tab, 1.0, True)
        assert np.array_equal(got, oracle.warp_batch(full["vic"], masks[:1], thetas[:1], tab))


def test_warp_batch_cifar_and_patchguard_bit_exact(oracle):
    for pg in (False, True):
        c = cifar_case(patchguard=pg)
        masks = np.concatenate([c["hull"][None], synth.synth_patch_masks("cf", 2, c["hull"])])
        thetas = ((c["tar_s"] - c["vic_s"])[None] * masks).astype(np.float32)
        got = _hip.warp_batch(c["vic"], masks, thetas, c["tab"], 1.0, pg)
        ref = oracle.warp_batch(c["vic"], masks, thetas, c["tab"], normalize=pg)
        assert np.array_equal(got, ref)


def test_edge_cases(oracle):
    case = gtsrb_case()
    vic = case["vic"]
    z = np.zeros((1, 3, 32, 32), np.float32)
    # empty mask: the composite is the victim image itself
    got = _hip.warp_batch(vic, z, z, case["tab"][:8], 1.0, True)
    assert np.array_equal(got, oracle.warp_batch(vic, z, z, case["tab"][:8]))
    # lbd = 0 (final re-evaluation, main.py:129) and saturating theta
    ones = np.ones((1, 3, 32, 32), np.float32)
    for lbd, th in ((0.0, z), (1.0, ones * 3.0), (1.0, -ones * 3.0)):
        got = _hip.warp_batch(vic, ones, th, case["tab"][:8], lbd, True)
        assert np.array_equal(got, oracle.warp_batch(vic, ones, th, case["tab"][:8], lbd=lbd))
    # zero candidates
    out = _hip.warp_batch(vic, np.zeros((0, 3, 32, 32), np.float32), np.zeros((0, 3, 32, 32), np.float32), case["tab"][:4],
                          1.0, True)
    assert out.shape == (0, 4, 3, 32, 32)
    # bad blur kernel is rejected with a message
    bad = case["tab"][:2].copy()
    bad["blur"] = 4
    with pytest.raises(_hip.GraphiteHipError, match="blur"):
        _hip.XformTable(bad)


def test_c_abi_error_codes(gtsrb_model):
    """Errors come back as negative codes with a message; nothing is launched and nothing throws across the ABI."""
    lib, ctx = _hip.load_library(), _hip.ctx()
    case = gtsrb_case()
    masks, thetas = gtsrb_candidates(case, 2)
    i, m, t = _hip.to_dev(case["vic"]), _hip.to_dev(masks), _hip.to_dev(thetas)
    xf = _hip.xform_table(case["tab"][:10])
    hits = torch.zeros(2, dtype=torch.int32, device="cuda")
    h = gtsrb_model.hip_model._h
    P = _hip._p

    def call(model=h, img=P(i), C=2, begin=0, count=20, hits_p=P(hits), c=ctx):
        return lib.gr_tr_hits(c, model, img, 244, 244, P(m), P(t), C, 32, 32, xf._h, 1.0, 1, 1, begin, count, hits_p, None, None,
                              None)

    assert call() == 0
    assert call(model=None) == -1 and b"bad arguments" in lib.gr_last_error(ctx)            # GR_EINVAL
    assert call(img=None) == -1
    assert call(hits_p=None) == -1
    assert call(begin=5, count=16) == -1 and b"slice out of range" in lib.gr_last_error(ctx)
    assert call(begin=-1) == -1
    assert call(C=-3) == -1
    assert call(c=None) == -1                                                                 # no context: code only
    assert call(begin=20, count=0) == 0                                                       # empty slice is fine
    assert lib.gr_ctx_set_chunk(ctx, 7) == -1 and b"chunk" in lib.gr_last_error(ctx)
    assert lib.gr_model_forward(ctx, h, None, 4, None, None) == -1
    torch.cuda.synchronize()
    # the context still works after the failures
    got, _, _ = _hip.tr_hits_dev(gtsrb_model.hip_model, i, m, t, case["tab"][:10], 1.0, True, 1)
    assert got.shape == (2,) and int(got.max()) <= 10


def test_reference_signatures(oracle):
    """get_transformed_images / apply_transformation / convert2Network return what the reference returns."""
    case = gtsrb_case()
    masks, thetas = gtsrb_candidates(case, 1)
    img, m, th = torch.from_numpy(case["vic"]), torch.from_numpy(masks[0]), torch.from_numpy(thetas[0])
    xf = case["xf"][:10]
    imgs = T.get_transformed_images(img, m, xf, 1.0, th, case["pt_file"])
    assert isinstance(imgs, list) and len(imgs) == 10 and imgs[0].shape == (3, 32, 32) and imgs[0].device.type == "cpu"
    ref = oracle.warp_batch(case["vic"], masks, thetas, case["tab"][:10])[0]
    assert np.array_equal(torch.stack(imgs).numpy(), ref)
    # the per-transform entry gives the same image
    att, pert, ml = T.add_noise(img, m, 1.0, th, True)
    one = T.apply_transformation(img, ml, pert, *xf[3][:7], case["pt_file"], 32, xf[3][7], xf[3][8], xf[3][9])
    assert np.array_equal(one.numpy(), ref[3])
    # no transforms: [convert2Network(att)]
    single = T.get_transformed_images(img, m, [], 1.0, th, case["pt_file"])
    comb, _, _ = oracle.add_noise(case["vic"], masks[0], thetas[0])
    want01 = np.clip(oracle.resize_linear(comb, 32, 32), 0, 1)
    assert len(single) == 1 and np.array_equal(single[0].numpy(), want01 - np.float32(0.5))
    hwc = np.ascontiguousarray(np.transpose(comb, (1, 2, 0)))
    assert np.array_equal(T.convert2Network(hwc, False, model_type="CIFAR").numpy(), want01)


# ------------------------------------------------------------------------------ classifiers
def test_gtsrbnet_logits_vs_oracle_and_reference(oracle, gtsrb_model, gtsrb_tensors):
    g = np.load(os.path.join(GOLD, "nets.npz"))
    x = synth.synth_images("gtsrb-golden", 64)
    lg = gtsrb_model.forward(torch.from_numpy(x)).cpu().numpy()
    ref = g["gtsrb_logits"]
    assert np.abs(lg - ref).max() <= REL_TOL * np.abs(ref).max()          # vs the reference's PyTorch forward
    assert np.array_equal(lg.argmax(1), g["gtsrb_preds"])
    orc = oracle.gtsrb_forward(gtsrb_tensors, x)
    assert np.abs(lg - orc).max() <= REL_TOL * np.abs(orc).max()
    assert np.array_equal(lg, orc), f"max |hip-oracle| = {np.abs(lg - orc).max()}"   # same fmaf chains
    p = gtsrb_model.predict(torch.from_numpy(x))
    assert p.dtype == torch.int64 and np.array_equal(p.cpu().numpy(), g["gtsrb_preds"])
    assert gtsrb_model.predict(torch.from_numpy(x[3])) == int(g["gtsrb_single_pred"])


def test_gtsrbnet_ragged_batches(oracle, gtsrb_model, gtsrb_tensors):
    x = synth.synth_images("gtsrb-ragged", 333)
    orc = oracle.gtsrb_forward(gtsrb_tensors, x)
    for n in (1, 3, 130, 333):
        lg = gtsrb_model.forward(torch.from_numpy(x[:n])).cpu().numpy()
        assert np.array_equal(lg, orc[:n])
    _hip.set_chunk(100)                                  # force several internal chunks
    try:
        assert np.array_equal(gtsrb_model.forward(torch.from_numpy(x)).cpu().numpy(), orc)
    finally:
        _hip.set_chunk(8192)
    # linear layers: launches of <= 640 rows run on 16x16x4 MFMA tiles (fc_small.h), larger ones on 32x32x2 tiles --
    # the same k-ascending chain per logit, so a batch and its slices agree bit for bit on either side of the switch
    xl = synth.synth_images("gtsrb-ragged-l", 1500)
    orc_l = oracle.gtsrb_forward(gtsrb_tensors, xl[:700])
    big = gtsrb_model.forward(torch.from_numpy(xl)).cpu().numpy()
    assert np.array_equal(big[:700], orc_l)
    for n in (640, 641, 700):
        _hip.profile_reset(); _hip.profile_enable(True)
        lg = gtsrb_model.forward(torch.from_numpy(xl[:n])).cpu().numpy()
        _hip.profile_enable(False)
        assert np.array_equal(lg, orc_l[:n])
        assert any("k_fc_small" in k for k in _hip.profile_read()) == (n <= 640)
    # predict() normalisation quirk (GTSRBNet.py:55-57)
    x01 = (x[:16] + np.float32(0.5)).astype(np.float32)
    renorm = (np.clip(x01, 0, 1) - np.float32(0.5)).astype(np.float32)   # what predict() feeds the net for out-of-range input
    want = oracle.gtsrb_forward(gtsrb_tensors, renorm).argmax(1)
    assert np.array_equal(gtsrb_model.predict(torch.from_numpy(x01)).cpu().numpy(), want)


def test_wrn_small_vs_oracle(oracle):
    sd = synth.wrn_state_dict("wrn-small", depth=10, widen=2, num_classes=10)
    m = models.WideResNet(depth=10, num_classes=10, widen_factor=2)
    m.load_state_dict(sd)
    x = synth.synth_images("wrn-small", 37, 0.0, 1.0)
    lg = m.forward(torch.from_numpy(x)).cpu().numpy()
    orc = oracle.wrn_forward(weights.wrn_tensor_list(sd, 10, 2), 10, 2, 10, x)
    assert np.abs(lg - orc).max() <= REL_TOL * np.abs(orc).max()
    assert np.array_equal(lg, orc), f"max |hip-oracle| = {np.abs(lg - orc).max()}"
    # depth 16 exercises the identity-shortcut (equalInOut) blocks
    sd = synth.wrn_state_dict("wrn-d16", depth=16, widen=1, num_classes=7)
    m = models.WideResNet(depth=16, num_classes=7, widen_factor=1)
    m.load_state_dict(sd)
    lg = m.forward(torch.from_numpy(x[:9])).cpu().numpy()
    orc = oracle.wrn_forward(weights.wrn_tensor_list(sd, 16, 1), 16, 1, 7, x[:9])
    assert np.array_equal(lg, orc), f"max |hip-oracle| = {np.abs(lg - orc).max()}"


def test_wrn_34_10_vs_reference_golden(oracle):
    g = np.load(os.path.join(GOLD, "nets.npz"))
    sd = synth.wrn_state_dict()
    m = models.WideResNet()
    m.load_state_dict(sd)
    x = synth.synth_images("wrn-golden", 8, 0.0, 1.0)
    lg = m.forward(torch.from_numpy(x)).cpu().numpy()
    ref = g["wrn_logits"]
    assert np.abs(lg - ref).max() <= REL_TOL * np.abs(ref).max()
    assert np.array_equal(lg.argmax(1), g["wrn_preds"])
    orc = oracle.wrn_forward(weights.wrn_tensor_list(sd), 34, 10, 10, x[:2])
    assert np.array_equal(lg[:2], orc), f"max |hip-oracle| = {np.abs(lg[:2] - orc).max()}"
    assert np.array_equal(m.predict(torch.from_numpy(x)).cpu().numpy(), g["wrn_preds"])
    # ragged batches through the halo-tile kernel (4 images per workgroup in the 8x8 layers): same rows, bit for bit
    for n in (1, 5, 7):
        assert np.array_equal(m.forward(torch.from_numpy(x[:n])).cpu().numpy(), lg[:n])


# ------------------------------------------------------------------------------ fused TR evaluation
def test_tr_hits_config1(oracle, gtsrb_model, gtsrb_tensors):
    """Config 1: stop -> speed-limit-30, hull mask (+ sub-masks), 100 seed-42 transforms."""
    case = gtsrb_case()
    masks, thetas = gtsrb_candidates(case, 6)
    imgs = oracle.warp_batch(case["vic"], masks, thetas, case["tab"]).reshape(-1, 3, 32, 32)
    lg = oracle.gtsrb_forward(gtsrb_tensors, imgs)
    preds_ref = lg.argmax(1)
    target = int(np.bincount(preds_ref).argmax())
    _, hits_ref = oracle.argmax_hits(lg, 100, target)
    hits, preds = engine.tr_hits(gtsrb_model, case["vic"], masks, thetas, case["tab"], target, return_preds=True)
    assert np.array_equal(preds.cpu().numpy().reshape(-1), preds_ref)
    assert np.array_equal(hits.cpu().numpy(), hits_ref)
    assert 0 < hits_ref.sum() < 600 and len(set(hits_ref.tolist())) > 1        # a non-trivial case
    # the reference's two-call form gives the same numbers
    m0, t0 = torch.from_numpy(masks[0]), torch.from_numpy(thetas[0])
    xi = T.get_transformed_images(torch.from_numpy(case["vic"]), m0, case["xf"], 1.0, t0, case["pt_file"])
    err, n = utils.run_predictions(gtsrb_model, xi, 14, target)
    assert n == 100 and err == 1.0 - hits_ref[0] / 100.0
    erru, _ = utils.run_predictions(gtsrb_model, xi, target)
    assert erru == 1.0 - (preds_ref[:100] != target).sum() / 100.0
    # every target label, one candidate
    for tgt in (0, 1, 14, 42):
        h = engine.tr_hits(gtsrb_model, case["vic"], masks[:1], thetas[:1], case["tab"], tgt)
        assert int(h[0]) == int((preds_ref[:100] == tgt).sum())


def test_tr_hits_chunking_and_slices(oracle, gtsrb_model, gtsrb_tensors):
    case = gtsrb_case()
    masks, thetas = gtsrb_candidates(case, 5)
    tab = case["tab"][:37]                               # T not a multiple of anything
    imgs = oracle.warp_batch(case["vic"], masks, thetas, tab).reshape(-1, 3, 32, 32)
    lg_ref = oracle.gtsrb_forward(gtsrb_tensors, imgs)
    preds_ref = lg_ref.argmax(1)
    target = int(np.bincount(preds_ref).argmax())
    _, hits_ref = oracle.argmax_hits(lg_ref, 37, target)
    hm = gtsrb_model.hip_model
    _hip.set_chunk(64)                                   # chunks straddle candidates
    try:
        hits, preds, logits = _hip.tr_hits_dev(hm, case["vic"], masks, thetas, tab, 1.0, True, target, want_preds=True,
                                               want_logits=True)
        assert np.array_equal(hits.cpu().numpy(), hits_ref)
        assert np.array_equal(preds.cpu().numpy(), preds_ref)
        assert np.array_equal(logits.cpu().numpy(), lg_ref)
        # slices (what each rank of a sharded run computes) sum to the whole
        total = 5 * 37
        acc = np.zeros(5, np.int64)
        for r in range(3):
            b, c = engine.shard_range(total, r, 3)
            h, p, _ = _hip.tr_hits_dev(hm, case["vic"], masks, thetas, tab, 1.0, True, target, b, c, want_preds=True)
            acc += h.cpu().numpy()
            p = p.cpu().numpy()
            assert np.array_equal(p[b:b + c], preds_ref[b:b + c]) and (p[:b] == -1).all() and (p[b + c:] == -1).all()
        assert np.array_equal(acc, hits_ref)
        h, _, _ = _hip.tr_hits_dev(hm, case["vic"], masks, thetas, tab, 1.0, True, target, 10, 0)
        assert int(h.sum()) == 0
    finally:
        _hip.set_chunk(8192)


def test_tr_hits_cifar_wrn_small(oracle):
    """Config 3 shape (RGF batch: 10 directions) on a small WRN so the oracle finishes in seconds."""
    c = cifar_case(num_xforms=20)
    sd = synth.wrn_state_dict("wrn-small", depth=10, widen=2, num_classes=10)
    m = models.WideResNet(depth=10, num_classes=10, widen_factor=2)
    m.load_state_dict(sd)
    base = ((c["tar_s"] - c["vic_s"]) * c["hull"]).astype(np.float32)
    dirs = synth.hash_uniform("rgf", (10, 3, 32, 32), -1, 1)
    dirs /= np.sqrt((dirs.reshape(10, -1) ** 2).sum(1))[:, None, None, None]
    thetas = (base[None] + dirs).astype(np.float32)
    masks = np.repeat(c["hull"][None], 10, 0)
    imgs = oracle.warp_batch(c["vic"], masks, thetas, c["tab"], normalize=False).reshape(-1, 3, 32, 32)
    lg = oracle.wrn_forward(weights.wrn_tensor_list(sd, 10, 2), 10, 2, 10, imgs)
    target = int(np.bincount(lg.argmax(1)).argmax())
    _, hits_ref = oracle.argmax_hits(lg, 20, target)
    hits, preds = engine.tr_hits(m, c["vic"], masks, thetas, c["tab"], target, pt_file=None, model_type="CIFAR",
                                 return_preds=True)
    assert np.array_equal(preds.cpu().numpy().reshape(-1), lg.argmax(1))
    assert np.array_equal(hits.cpu().numpy(), hits_ref)


def test_full_size_properties(gtsrb_model):
    """Size-independent checks at benchmark scale (C x T = 64 x 1000): permutation equivariance over
    candidates, slice additivity, determinism."""
    case = gtsrb_case(num_xforms=1000)
    masks, thetas = gtsrb_candidates(case, 64, tag="full")
    h1 = engine.tr_hits(gtsrb_model, case["vic"], masks, thetas, case["tab"], 1).cpu().numpy()
    perm = np.random.RandomState(0).permutation(64)
    h2 = engine.tr_hits(gtsrb_model, case["vic"], masks[perm], thetas[perm], case["tab"], 1).cpu().numpy()
    assert np.array_equal(h1[perm], h2)
    assert (h1 >= 0).all() and (h1 <= 1000).all()
    hm = gtsrb_model.hip_model
    acc = np.zeros(64, np.int64)
    for r in range(8):
        b, c = engine.shard_range(64000, r, 8)
        acc += _hip.tr_hits_dev(hm, case["vic"], masks, thetas, case["tab"], 1.0, True, 1, b, c)[0].cpu().numpy()
    assert np.array_equal(acc, h1)
    # sum over all 43 targets of hits == T for each candidate (every image gets exactly one label)
    tot = np.zeros(4, np.int64)
    for tgt in range(43):
        tot += engine.tr_hits(gtsrb_model, case["vic"], masks[:4], thetas[:4], case["tab"][:100], tgt).cpu().numpy()
    assert (tot == 100).all()
