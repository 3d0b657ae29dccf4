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
