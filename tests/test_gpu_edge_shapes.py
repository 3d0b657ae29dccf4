"""Edge shapes of the fused path against the oracle, bit for bit: frames that are neither 244 x 244 nor square (odd
sizes, tiny, larger than a sign), one transform, many candidates with one transform each, chunks that cut candidates in
the middle, a workspace limit small enough to force the smallest chunks, slices of the (candidate x transform) index."""
import numpy as np
import pytest

import oracle_lib as O
from graphite_amd import _hip, engine, models, synth, transforms as T, weights

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def net():
    sd = synth.gtsrb_state_dict(readout="toy3")
    m = models.GTSRBNet()
    m.load_state_dict(sd)
    return m, weights.gtsrb_tensor_list(sd)


def _case(tag, H, W, C, Tn, seed, baseline=False):
    img = synth.hash_uniform(f"edge-{tag}/img", (3, H, W), 0.0, 1.0)
    img = (np.floor(img * 256).clip(0, 255) / 255.0).astype(np.float32)
    masks = synth.synth_patch_masks(f"edge-{tag}", C, np.ones((3, 32, 32), np.float32), p_keep=0.5)
    thetas = (synth.hash_uniform(f"edge-{tag}/theta", (C, 3, 32, 32), -0.5, 0.5) * masks).astype(np.float32)
    np.random.seed(seed)
    xf = T.get_transform_params(Tn, "GTSRB", baseline=baseline)
    return img, masks, thetas, T.build_xform_table(xf, W, H, None)


def _check(net, img, masks, thetas, tab, target=14, lbd=1.0):
    m, tl = net
    Tn = len(tab)
    ref = O.warp_batch(img, masks, thetas, tab, lbd=lbd)
    assert np.array_equal(_hip.warp_batch(img, masks, thetas, tab, lbd, True), ref)
    lg = O.gtsrb_forward(tl, ref.reshape(-1, 3, 32, 32))
    _, h_ref = O.argmax_hits(lg, Tn, target)
    hits, preds = engine.tr_hits(m, img, masks, thetas, tab, target, lbd=lbd, return_preds=True)
    assert np.array_equal(hits.cpu().numpy(), h_ref)
    assert np.array_equal(preds.cpu().numpy().reshape(-1), lg.argmax(1))
    return h_ref


@pytest.mark.parametrize("H,W", [(61, 47), (47, 61), (33, 33), (97, 244), (300, 211), (512, 384)])
def test_odd_frames(net, H, W):
    img, masks, thetas, tab = _case(f"{H}x{W}", H, W, 2, 9, seed=H * 1000 + W)
    _check(net, img, masks, thetas, tab)


def test_one_transform_and_many_candidates(net):
    img, masks, thetas, tab = _case("one", 244, 244, 1, 1, seed=5)
    _check(net, img, masks, thetas, tab)
    img, masks, thetas, tab = _case("many", 120, 120, 257, 1, seed=6)     # every chunk holds many candidates' composites
    _check(net, img, masks, thetas, tab)


def test_chunks_that_cut_candidates_and_a_tiny_workspace(net):
    img, masks, thetas, tab = _case("chunks", 244, 244, 5, 37, seed=7, baseline=True)
    try:
        _hip.set_chunk(64)                       # 185 images in chunks of 64: candidates are cut mid-way
        h64 = _check(net, img, masks, thetas, tab)
        _hip.set_chunk(8192)
        _hip.set_workspace_limit(64 << 20)       # the smallest limit the library takes: chunks of a few hundred images
        hws = _check(net, img, masks, thetas, tab)
    finally:
        _hip.set_chunk(8192)
        _hip.set_workspace_limit(8 << 30)
    href = _check(net, img, masks, thetas, tab)
    assert np.array_equal(h64, href) and np.array_equal(hws, href)


def test_slices_of_the_index_add_up(net):
    m, _ = net
    img, masks, thetas, tab = _case("slices", 244, 244, 4, 25, seed=8)
    xf = _hip.xform_table(tab)
    img_d, m_d, t_d = _hip.to_dev(img), _hip.to_dev(masks), _hip.to_dev(thetas)
    full, _, _ = _hip.tr_hits_dev(m.hip_model, img_d, m_d, t_d, xf, 1.0, True, 14, 0, 100)
    parts = None
    for b, c in ((0, 1), (1, 24), (25, 0), (25, 50), (75, 13), (88, 12)):     # cuts inside candidates, an empty slice
        h, _, _ = _hip.tr_hits_dev(m.hip_model, img_d, m_d, t_d, xf, 1.0, True, 14, b, c)
        parts = h.clone() if parts is None else parts + h
    assert np.array_equal(parts.cpu().numpy(), full.cpu().numpy())
