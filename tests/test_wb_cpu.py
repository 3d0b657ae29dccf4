"""CPU checks of the whitebox transform's host side and of its torch restatement (oracle/wb_reference.py)."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import wb_reference as R                                                   # noqa: E402
from cases import PT14, gtsrb_case                                          # noqa: E402
from graphite_amd import transforms as T                                    # noqa: E402


def test_wb_table_matches_reference_restatement():
    c = gtsrb_case(num_xforms=24)
    tab = T.build_wb_table(c["xf"], 244, 244, PT14)
    for t in range(len(tab)):
        xf = c["tab"][t]
        A = R.src_norm_trans_dst_norm(xf["M"].reshape(3, 3), (244, 244), (int(xf["crop_y"]), int(xf["crop_x"])))
        assert np.array_equal(tab[t]["A"], A.numpy().reshape(9))
        assert tab[t]["crop_x"] == xf["crop_x"] and tab[t]["crop_y"] == xf["crop_y"] and tab[t]["blur"] == xf["blur"]
        assert tab[t]["gamma"] == np.float32(c["xf"][t][2])


def test_reference_identity_warp_is_a_plain_resize():
    """M = identity with dsize = frame: kornia's align_corners=True convention maps pixel centres onto pixel centres, so
    the transform reduces to where(mask) -> pow -> F.interpolate."""
    g = torch.Generator().manual_seed(0)
    orig = torch.rand(1, 3, 64, 64, generator=g) * 0.9 + 0.05
    att = torch.rand(1, 3, 64, 64, generator=g) * 0.9 + 0.05
    mask = torch.zeros(3, 64, 64)
    mask[:, 10:50, 20:60] = 1
    out = R.transform_wb(orig, att, mask, np.eye(3), 64, 64, 1.7, 0)
    comp = torch.where(mask > 0.5, att, orig)
    want = F.interpolate(torch.pow(comp, 1.7), size=(32, 32), mode="bilinear", align_corners=False)
    assert float((out - want).abs().max()) < 2e-6


def test_reference_blur_kernel_and_mask_semantics():
    k = R.blur_kernel(5)
    assert k.shape == (5, 5) and abs(k.sum() - 1.0) < 1e-15 and k[2, 2] == 0.375 ** 2
    # outside the mask the original shows through, untouched by the blur of the perturbation
    orig = torch.full((1, 3, 32, 32), 0.25)
    att = torch.full((1, 3, 32, 32), 0.75)
    mask = torch.zeros(3, 32, 32)
    mask[:, 8:24, 8:24] = 1
    out = R.transform_wb(orig, att, mask, np.eye(3), 32, 32, 1.0, 7)
    assert float((out[0, :, :4, :4] - 0.25).abs().max()) < 1e-6
    assert float((out[0, :, 15, 15] - 0.75).abs().max()) < 1e-6           # deep inside: full kernel support is masked
    assert float(out[0, 0, 8, 8]) < 0.75                                    # at the mask corner the zero-padded perturbation leaks in


def test_convert2networkwb_matches_interpolate():
    x = torch.rand(2, 3, 77, 77) * 1.4 - 0.2
    y = T.convert2NetworkWB(x)
    assert y.shape == (2, 3, 32, 32) and float(y.min()) >= 0 and float(y.max()) <= 1
    assert torch.equal(T.convert2NetworkWB(x[0]), y[0])


def _literal_pixel_loop(mask, grad, count):
    """whitebox/whitebox_attack.py:223-235 written out literally (CPU tensors)."""
    ref_mask, fg = mask.clone(), grad.clone()
    for _ in range(count):
        fg = ref_mask * fg
        s = torch.sum(fg, dim=0)
        s = torch.where(s == 0, torch.tensor(999999).float(), s)
        s[torch.isnan(s)] = 999999
        idx = (s == torch.min(s)).nonzero()[0]
        for c in range(3):
            ref_mask[(c,) + tuple(idx)] = 0
    return ref_mask


def test_remove_pixels_closed_form_equals_the_reference_loop():
    """Includes the case the round-2 review found: count > live pixels.  Once every entry is the 999999 sentinel the loop
    keeps hitting flat index 0, so pixel (0, 0) loses its mask bits -- whether it was live, a zero-gradient masked pixel
    or a NaN one -- and no other sentinel pixel does."""
    from graphite_amd.whitebox_attack import _remove_pixels
    g = torch.Generator().manual_seed(3)
    for case in range(8):
        mask = (torch.rand(1, 12, 15, generator=g) > 0.4).float().repeat(3, 1, 1)
        grad = torch.randn(3, 12, 15, generator=g)
        grad[:, 2, 3] = 0.0                                   # a live pixel with a zero sum
        grad[0, 5, 5] = float("nan")
        if case >= 3:
            mask[:, 0, 0] = 1.0                               # (0, 0) inside the mask ...
            mask[:, 0, 1] = 1.0
            grad[:, 0, 1] = 0.0                               # ... and a masked zero-gradient pixel right after it
        if case in (4, 6):
            grad[:, 0, 0] = 0.0                               # (0, 0) masked with a zero gradient sum: a sentinel from the start
        if case in (5, 7):
            grad[1, 0, 0] = float("nan")
        count = (7, 60, 400, 400, 400, 400, 1, 123)[case]
        want = _literal_pixel_loop(mask, grad, count)
        got = mask.clone()
        _remove_pixels(got, grad, count)
        assert torch.equal(got, want), case
