"""SURVEY 8f row 4, pixel half: frames of any aspect ratio (perspective_transform.py:39-44 gives a non-square dsize) and
convert2Network's net_size != 32 branch (transforms.py:31-45: short side -> net_size, aspect kept, clamp only).
The geometry (M, dsize) of these frames is pinned by the reference itself (homography.npz 'nonsquare',
tests/test_host_golden.py); here the size rule is checked against the reference's formula and the pixels of the HIP path
against the oracle, bit for bit.  The `alpr` classifier that consumes such images is an external binary and stays out."""
import os

import numpy as np
import pytest
import torch

from cases import INP, PT14
from graphite_amd import _hip, synth, transforms as T


def _ref_size(crop_x, crop_y, net):
    """transforms.py:35-40 verbatim in Python floats (img.shape = (crop_y, crop_x))."""
    if net == 32:
        return 32, 32
    if crop_y > crop_x:
        return net, int(round(net / crop_x * crop_y))
    return int(round(net / crop_y * crop_x)), net


def test_net_size_rule_matches_the_reference_formula(oracle):
    lib = _hip.load_library()                               # host helper of the library: no GPU needed
    rng = np.random.RandomState(3)
    cases = [(64, 64, 32), (100, 50, 32), (50, 100, 24), (100, 50, 24), (97, 97, 48)]
    cases += [(int(a), int(b), int(n)) for a, b, n in zip(rng.randint(20, 400, 400), rng.randint(20, 400, 400),
                                                         rng.choice([16, 20, 24, 48, 64, 100], 400))]
    for cx, cy, net in cases:
        want = _ref_size(cx, cy, net)
        assert oracle.net_size(cx, cy, net) == want
        assert _hip.net_size(cx, cy, net) == want
    assert lib.gr_net_size(0, 5, 24, None, None) != 0


def _frame_case(w, h, n_xf, seed):
    img = synth.synth_victim(f"frame-{w}x{h}", max(w, h))[:, :h, :w]
    img = np.ascontiguousarray(img)
    np.random.seed(seed)
    xf = T.get_transform_params(n_xf, "GTSRB")
    for pt in (PT14, None):
        yield img, xf, T.build_xform_table(xf, w, h, pt), pt


@pytest.mark.gpu
@pytest.mark.parametrize("w,h", [(300, 200), (200, 300), (244, 244)])
def test_variable_size_branch_bit_exact(oracle, w, h):
    for img, xf, tab, pt in _frame_case(w, h, 14, 7):
        if w != h:
            assert (tab["crop_x"] != tab["crop_y"]).all()            # non-square dsize, as perspective_transform.py:39-44
        masks = synth.synth_patch_masks("ns", 2, np.ones((3, 32, 32), np.float32), p_keep=0.6)
        thetas = (synth.hash_uniform("ns-theta", (2, 3, 32, 32), -0.4, 0.4) * masks).astype(np.float32)
        for net in (48, 20):
            got = _hip.warp_batch_var(img, masks, thetas, tab, 1.0, net)
            ref = oracle.warp_batch_var(img, masks, thetas, tab, net)
            assert len(got) == len(ref) == 2 * len(tab)
            for i, (g, r) in enumerate(zip(got, ref)):
                t = tab[i % len(tab)]
                nx, ny = _ref_size(int(t["crop_x"]), int(t["crop_y"]), net)
                assert g.shape == r.shape == (3, ny, nx) and min(nx, ny) == net
                assert np.array_equal(g, r), f"frame {w}x{h} net {net} image {i}: {(g != r).sum()} values differ"
                assert g.min() >= 0.0 and g.max() <= 1.0 and g.std() > 0.01   # clamp only: never normalised to [-0.5, 0.5]
        # net_size 32 on a non-square frame: still 32 x 32 (transforms.py:19-29), through the fused kernel
        got32 = _hip.warp_batch(img, masks, thetas, tab, 1.0, True)
        assert np.array_equal(got32, oracle.warp_batch(img, masks, thetas, tab))
        # the reference-signature entry points
        imgs = T.get_transformed_images(torch.from_numpy(img), torch.from_numpy(masks[0]), xf, 1.0, torch.from_numpy(thetas[0]),
                                        pt, net_size=48)
        ref = oracle.warp_batch_var(img, masks[:1], thetas[:1], tab, 48)
        assert len(imgs) == len(xf) and all(np.array_equal(a.numpy(), b) for a, b in zip(imgs, ref))
        att, pert, ml = T.add_noise(torch.from_numpy(img), torch.from_numpy(masks[0]), 1.0, torch.from_numpy(thetas[0]), True)
        one = T.apply_transformation(torch.from_numpy(img), ml, pert, *xf[5][:7], pt, 48, xf[5][7], xf[5][8], xf[5][9])
        assert np.array_equal(one.numpy(), ref[5])


@pytest.mark.gpu
def test_convert2network_variable_size(oracle):
    for (h, w) in ((60, 140), (140, 60), (77, 77), (96, 48)):
        x = synth.hash_uniform(f"c2n-{h}x{w}", (3, h, w), -0.2, 1.2)
        for net in (24, 48):
            nx, ny = _ref_size(w, h, net)
            got = T.convert2Network(torch.from_numpy(x), True, net_size=net).numpy()
            want = np.clip(oracle.resize_linear(x, ny, nx), 0, 1)
            assert got.shape == (3, ny, nx) and np.array_equal(got, want)
