"""a13 on the GPU: the PatchGuard robust classifier through the C ABI against the CPU oracle (bit for bit) and against the
reference's own torch lines for the pre-processing (1e-5).  BagNet and masking_defense are third-party code restated from
the publications (include/graphite_hip.h): HIP == oracle is proven here, oracle == PatchGuard's source is PARITY UNPINNED."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import patchguard_reference as PR                           # noqa: E402
import oracle_lib as O                                      # noqa: E402
from cases import cifar_case                                # noqa: E402
from graphite_amd import _hip, engine, synth, weights       # noqa: E402
from graphite_amd.patchguard.patchguard_predict import PatchGuard, PatchGuardNoDefense     # noqa: E402

pytestmark = pytest.mark.gpu

SMALL = dict(layers=(1, 1, 1, 1), strides=(2, 2, 2, 1), kernel3=(1, 1, 1, 0))
B33 = dict(layers=(2, 2, 2, 2), strides=(2, 2, 1, 1), kernel3=(1, 1, 1, 1))


@pytest.mark.parametrize("size", [192, 64, 50])
def test_preprocess_kernel_bit_exact_vs_oracle_and_1e5_vs_torch(size):
    x = synth.synth_images("pg-pre", 5)
    x[0, :, :2, :] = -0.5; x[0, :, -1, :] = 0.5
    got = _hip.pg_preprocess(x, size).cpu().numpy()
    assert np.array_equal(got, O.pg_preprocess(x, size))
    ref = PR.preprocess(torch.from_numpy(x), size).numpy()                    # the reference's lines 42-44, literally
    assert np.abs(got - ref).max() <= 1e-5


def test_masking_kernel_bit_exact_vs_oracle():
    rng = np.random.RandomState(5)
    for (n, h, w, ncls, win) in ((40, 22, 22, 10, 6), (7, 6, 6, 10, 6), (5, 9, 13, 43, 4), (3, 1, 1, 3, 1)):
        local = (rng.randn(n, h, w, ncls) * 3).astype(np.float32)
        local[0] = -np.abs(local[0])
        if h >= 9:
            local[1, 1:7, 2:8, ncls - 3] += 40.0
        local[2] = 0.0
        for thres in (0.0, 0.25):
            s_ref, p_ref = O.masking_defense(local, win, thres)
            s, p = _hip.pg_masking_defense(local, win, thres)
            assert np.array_equal(s.cpu().numpy(), s_ref) and np.array_equal(p.cpu().numpy(), p_ref), (n, h, w, ncls, thres)
    with pytest.raises(_hip.GraphiteHipError):
        _hip.pg_masking_defense(np.zeros((1, 4, 4, 10), np.float32), 5, 0.0)  # window larger than the map
    big = (rng.randn(2, 64, 64, 10) * 3).astype(np.float32)                    # 160 KB per image: the path that does not stage in LDS
    s_ref, p_ref = O.masking_defense(big, 6, 0.0)
    s, p = _hip.pg_masking_defense(big, 6, 0.0)
    assert np.array_equal(s.cpu().numpy(), s_ref) and np.array_equal(p.cpu().numpy(), p_ref)


@pytest.mark.parametrize("arch,size,n", [(SMALL, 64, 9), (B33, 48, 5), (synth.BAGNET17, 192, 3)])
def test_local_logits_bit_exact_vs_oracle(arch, size, n):
    calibrated = arch is synth.BAGNET17
    sd = synth.bagnet_state_dict("bagnet-v1" if calibrated else "bagnet-t", arch["layers"], arch["kernel3"], calibrated=calibrated)
    x = synth.synth_images("pg-net", n)
    m = PatchGuard(sd, size=size, **arch)
    got = m.local_logits(x).cpu().numpy()
    ref = O.bagnet_forward(weights.bagnet_tensor_list(sd, arch["layers"]), O.pg_preprocess(x, size), **arch)
    assert got.shape == ref.shape
    assert np.array_equal(got, ref), float(np.abs(got - ref).max())
    # and against the plain PyTorch modules of the same architecture (BatchNorm not folded): rounding only
    net = PR.BagNet(arch["layers"], arch["strides"], arch["kernel3"]).eval()
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    with torch.no_grad():
        t = net(PR.preprocess(torch.from_numpy(x), size)).numpy()
    assert np.abs(got - t).max() <= 1e-4 * np.abs(t).max()
    # predictions: defended (last-maximum rule) and undefended (first-maximum rule)
    win = min(6, ref.shape[1])
    assert m.predict(x).cpu().tolist() == O.masking_defense(ref, win, 0.0)[1].tolist()
    nd = PatchGuardNoDefense(sd, size=size, **arch)
    assert nd.predict(x).cpu().tolist() == O.pg_mean_logits(ref).argmax(1).tolist()
    assert isinstance(m.predict(x[0]), int)
    # forward(): the masked class evidence itself
    assert np.array_equal(m.hip_model.forward(x).cpu().numpy(), O.masking_defense(ref, win, 0.0)[0])


def test_duck_typed_local_logit_net_keeps_the_hip_ends():
    """PatchGuard(net=...) -- HIP pre-processing, any callable in the middle, HIP masking epilogue."""
    sd = synth.bagnet_state_dict("bagnet-t", SMALL["layers"], SMALL["kernel3"], calibrated=False)
    net = PR.BagNet(SMALL["layers"], SMALL["strides"], SMALL["kernel3"]).eval()
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    seen = {}

    def middle(z):
        seen["z"] = z
        with torch.no_grad():
            return net(z.cpu())

    x = synth.synth_images("pg-duck", 6)
    m = PatchGuard(net=middle, size=64)
    got = m.predict(x).cpu().tolist()
    assert seen["z"].is_cuda and tuple(seen["z"].shape) == (6, 3, 64, 64)
    with torch.no_grad():
        lo = net(torch.from_numpy(O.pg_preprocess(x, 64))).numpy()
    assert got == O.masking_defense(lo, 6, 0.0)[1].tolist()


def _pg_candidates(c, n):
    masks = np.concatenate([c["hull"][None], synth.synth_patch_masks("pg-cand", n - 1, c["hull"])])
    thetas = ((c["tar_s"] - c["vic_s"])[None] * masks).astype(np.float32)
    return masks, thetas


def test_config5_tr_hits_through_the_robust_classifier_vs_oracle():
    """BASELINE config 5: pg parameter table (baseline) -> warp -> PatchGuard (bicubic x6, BagNet-17, masking_defense) -> hit
    counts, fused on the GPU, against the oracle image by image."""
    c = cifar_case(num_xforms=8, patchguard=True, baseline=True)
    masks, thetas = _pg_candidates(c, 3)
    sd = synth.bagnet_state_dict()
    m = PatchGuard(sd)
    px = O.warp_batch(c["vic"], masks, thetas, c["tab"], normalize=True).reshape(-1, 3, 32, 32)
    lo = O.bagnet_forward(weights.bagnet_tensor_list(sd), O.pg_preprocess(px))
    scores, preds = O.masking_defense(lo, 6, 0.0)
    target = int(np.bincount(preds).argmax())
    want = (preds.reshape(3, 8) == target).sum(1)
    hits, got = engine.tr_hits(m, c["vic"], masks, thetas, c["tab"], target, pt_file=None, model_type='PATCHGUARD', return_preds=True)
    assert got.cpu().numpy().reshape(-1).tolist() == preds.tolist()
    assert hits.cpu().numpy().tolist() == want.tolist()
    # the reference's two-call form gives the same numbers (utils.py:60 on get_transformed_images' list)
    from graphite_amd.patchguard import pg_transforms
    from graphite_amd import utils
    imgs = pg_transforms.get_transformed_images(torch.from_numpy(c["vic"]), torch.from_numpy(masks[1]), c["xf"], 1.0,
                                                torch.from_numpy(thetas[1]), None)
    err, nq = utils.run_predictions(m, imgs, -1, target)
    assert nq == 8 and abs((1 - err) * 8 - want[1]) < 1e-9
    # undefended twin on the same images
    nd = PatchGuardNoDefense(sd)
    h2, p2 = engine.tr_hits(nd, c["vic"], masks, thetas, c["tab"], target, pt_file=None, model_type='PATCHGUARD', return_preds=True)
    assert p2.cpu().numpy().reshape(-1).tolist() == O.pg_mean_logits(lo).argmax(1).tolist()


def test_empty_and_single_batches():
    sd = synth.bagnet_state_dict("bagnet-t", SMALL["layers"], SMALL["kernel3"], calibrated=False)
    m = PatchGuard(sd, size=64, **SMALL)
    e = np.zeros((0, 3, 32, 32), np.float32)
    assert tuple(_hip.pg_preprocess(e, 64).shape) == (0, 3, 64, 64)
    assert tuple(m.local_logits(e).shape) == (0, 6, 6, 10)
    s, p = _hip.pg_masking_defense(np.zeros((0, 6, 6, 10), np.float32), 6, 0.0)
    assert s.shape[0] == 0 and p.shape[0] == 0
    x = synth.synth_images("pg-one", 3)
    one = m.local_logits(x[1:2]).cpu().numpy()
    assert np.array_equal(one[0], m.local_logits(x).cpu().numpy()[1])       # an image's result does not depend on its batch
    with pytest.raises(_hip.GraphiteHipError):
        _hip.pg_masking_defense(np.zeros((2, 6, 6, 10), np.float32), 0, 0.0)   # window of zero


def test_patchguard_model_rejects_what_it_cannot_do():
    m = PatchGuard(synth.bagnet_state_dict("bagnet-t", SMALL["layers"], SMALL["kernel3"], calibrated=False), size=64, **SMALL)
    with pytest.raises(_hip.GraphiteHipError):
        m.hip_model.input_grad(synth.synth_images("pg-g", 1), 0)
    with pytest.raises(_hip.GraphiteHipError):
        m.hip_model.pg_configure(True, 99, 0.0)
    with pytest.raises(_hip.GraphiteHipError):
        _hip.Model("bagnet", weights.bagnet_tensor_list(synth.bagnet_state_dict("bagnet-t", SMALL["layers"], SMALL["kernel3"],
                                                                               calibrated=False), SMALL["layers"]),
                   layers=(1, 1, 1, 2), strides=(2, 2, 2, 1), kernel3=(1, 1, 1, 0), upsample=64)     # tensor count mismatch


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_patchguard_bounded(seed):
    """6 randomised BagNet geometries / inputs / masking windows of tests/fuzz_patchguard.py."""
    import fuzz_patchguard
    ok, msg = fuzz_patchguard.run_case(seed)
    assert ok, msg


@pytest.mark.parametrize("arch,size,n", [(SMALL, 64, 6), (synth.BAGNET17, 192, 4)])
def test_bf16_throughput_mode_follows_its_restatement(arch, size, n):
    """GR_PREC_BF16 for the PatchGuard classifier: the stride-1 1x1 convs on bf16 operands (f32 accumulation).  NOT the parity
    path.  Against plain PyTorch with the same operand rounding and float64 accumulation the local logits agree to a few
    1e-3 of the largest logit (an activation on a bf16 rounding boundary rounds differently after a 1e-7 difference in
    accumulation order, as for the other nets: DESIGN.md 5b); against the fp32 path to the bf16 error itself; switching
    back restores the bit-exact fp32 results."""
    calibrated = arch is synth.BAGNET17
    sd = synth.bagnet_state_dict("bagnet-v1" if calibrated else "bagnet-t", arch["layers"], arch["kernel3"], calibrated=calibrated)
    x = synth.synth_images("pg-bf16", n)
    m = PatchGuard(sd, size=size, **arch)
    f32 = m.local_logits(x).cpu().numpy()
    m.set_precision("bf16")
    got = m.local_logits(x).cpu().numpy()
    z = PR.preprocess(torch.from_numpy(x), size)
    with torch.no_grad():
        ref16 = PR.bagnet_local_logits_folded(sd, z, bf16=True, **arch).numpy()
        ref32 = PR.bagnet_local_logits_folded(sd, z, bf16=False, **arch).numpy()
    scale = np.abs(ref32).max()
    assert np.abs(f32 - ref32).max() <= 1e-4 * scale                       # the folded torch forward == the fp32 path
    e_ref = np.abs(got - ref16).max() / scale
    e_f32 = np.abs(got - f32).max() / scale
    print("bagnet bf16 mode: max |hip - restatement| / max|logit| = %.2e, median %.2e; vs fp32 path %.2e" % (
        e_ref, np.median(np.abs(got - ref16)) / scale, e_f32))
    assert e_ref <= 3e-2 and np.median(np.abs(got - ref16)) / scale <= 5e-3        # measured 1.5e-2 / 2.0e-3 at 16 blocks (bagnet17), see the print
    assert 1e-5 < e_f32 <= 5e-2                                             # it IS a different arithmetic, and a close one
    m.set_precision("f32")
    assert np.array_equal(m.local_logits(x).cpu().numpy(), f32)
