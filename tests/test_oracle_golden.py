"""CPU oracle vs the reference's own PyTorch forward (golden logits) and internal consistency."""
import os

import numpy as np
import pytest

from cases import GOLD, gtsrb_case, gtsrb_candidates, cifar_case
from graphite_amd import synth, weights


@pytest.fixture(scope="module")
def nets():
    return np.load(os.path.join(GOLD, "nets.npz"))


def test_weight_checksums(nets):
    assert synth.state_dict_checksum(synth.gtsrb_state_dict()) == str(nets["gtsrb_checksum"])
    sdw = synth.wrn_state_dict()
    assert len(sdw) == int(nets["wrn_num_entries"]) == 252
    assert synth.state_dict_checksum(sdw) == str(nets["wrn_checksum"])


def test_gtsrbnet_oracle_matches_reference_logits(oracle, nets):
    sd = synth.gtsrb_state_dict()
    x = synth.synth_images("gtsrb-golden", 64)
    lg = oracle.gtsrb_forward(weights.gtsrb_tensor_list(sd), x)
    ref = nets["gtsrb_logits"]
    # north star: logits within 1e-4 relative, argmax identical
    assert np.abs(lg - ref).max() <= 1e-4 * np.abs(ref).max()
    assert np.array_equal(lg.argmax(1), nets["gtsrb_preds"])
    assert len(set(nets["gtsrb_preds"].tolist())) > 5          # calibrated: not a constant predictor
    preds, hits = oracle.argmax_hits(lg, 64, int(nets["gtsrb_preds"][0]))
    assert np.array_equal(preds, nets["gtsrb_preds"]) and hits[0] == (nets["gtsrb_preds"] == nets["gtsrb_preds"][0]).sum()


def test_wrn_oracle_matches_reference_logits(oracle, nets):
    sd = synth.wrn_state_dict()
    x = synth.synth_images("wrn-golden", 8, 0.0, 1.0)
    lg = oracle.wrn_forward(weights.wrn_tensor_list(sd), 34, 10, 10, x)
    ref = nets["wrn_logits"]
    assert np.abs(lg - ref).max() <= 1e-4 * np.abs(ref).max()
    assert np.array_equal(lg.argmax(1), nets["wrn_preds"])


def test_oracle_small_wrn_vs_torch(oracle):
    """A small WRN-10-2 against a plain PyTorch fp32 restatement of the block structure."""
    import torch
    import torch.nn.functional as F
    sd = synth.wrn_state_dict("wrn-small", depth=10, widen=2, num_classes=10)
    x = synth.synth_images("wrn-small", 4, 0.0, 1.0)
    lg = oracle.wrn_forward(weights.wrn_tensor_list(sd, 10, 2), 10, 2, 10, x)
    t = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}

    def bn(h, p):
        return F.batch_norm(h, t[p + ".running_mean"], t[p + ".running_var"], t[p + ".weight"], t[p + ".bias"], False, 0.0, 1e-5)

    h = F.conv2d(torch.from_numpy(x), t["conv1.weight"], padding=1)
    for g, stride in ((1, 1), (2, 2), (3, 2)):
        p = f"block{g}.layer.0"
        o = F.relu(bn(h, p + ".bn1"))
        u = F.relu(bn(F.conv2d(o, t[p + ".conv1.weight"], stride=stride, padding=1), p + ".bn2"))
        v = F.conv2d(u, t[p + ".conv2.weight"], padding=1)
        h = F.conv2d(o, t[p + ".convShortcut.weight"], stride=stride) + v
    h = F.relu(bn(h, "bn1"))
    ref = (F.avg_pool2d(h, 8).view(-1, 128) @ t["fc.weight"].T + t["fc.bias"]).numpy()
    assert np.abs(lg - ref).max() <= 1e-4 * np.abs(ref).max()


def test_oracle_resize_properties(oracle):
    x = synth.hash_uniform("rs", (3, 32, 32))
    assert np.array_equal(oracle.resize_linear(x, 32, 32), x)                 # equal size = copy
    c = np.full((3, 17, 23), 0.25, np.float32)
    up = oracle.resize_linear(c, 244, 244)
    assert np.abs(up - 0.25).max() < 1e-7                                      # constants survive
    up = oracle.resize_linear(x, 244, 244)
    assert up.min() >= x.min() - 1e-6 and up.max() <= x.max() + 1e-6
    # torch's bilinear (align_corners=False) is the same tap rule away from the borders
    import torch
    ref = torch.nn.functional.interpolate(torch.from_numpy(x)[None], size=(244, 244), mode="bilinear",
                                          align_corners=False)[0].numpy()
    assert np.abs(up - ref).max() < 1e-5


def test_oracle_blur_and_composite(oracle):
    case = gtsrb_case()
    masks, thetas = gtsrb_candidates(case, 1)
    comb, pert, ml = oracle.add_noise(case["vic"], masks[0], thetas[0])
    assert set(np.unique(ml)) <= {0.0, 1.0} and comb.min() >= 0 and comb.max() <= 1
    assert np.array_equal(pert, np.where(ml > 0.5, comb, 0))
    a0 = oracle.composite(case["vic"], pert, ml, 0)
    assert np.array_equal(a0, np.where(ml > 0.5, comb, case["vic"]))
    import scipy.ndimage as ndi
    for k, ker in ((3, [.25, .5, .25]), (5, [.0625, .25, .375, .25, .0625]),
                   (7, [.03125, .109375, .21875, .28125, .21875, .109375, .03125])):
        att = oracle.composite(case["vic"], pert, ml, k)
        ker = np.array(ker, np.float64)
        b = ndi.convolve1d(ndi.convolve1d(pert.astype(np.float64), ker, axis=2, mode="mirror"), ker, axis=1, mode="mirror")
        ref = np.clip(np.where(ml > 0.5, b, case["vic"]), 0, 1)
        assert np.abs(att - ref).max() < 1e-6
        assert np.array_equal(att[ml < 0.5], case["vic"][ml < 0.5])


def test_oracle_warp_identity_and_pipeline(oracle):
    case = gtsrb_case()
    att = case["vic"]
    xf = np.zeros(1, oracle.XFORM_DTYPE)
    xf[0]["M"] = np.eye(3).reshape(9)
    xf[0]["crop_x"] = xf[0]["crop_y"] = 244
    xf[0]["lut"] = np.arange(256, dtype=np.uint8)
    out, warped = oracle.warp_one(att, xf[0], normalize=False, want_warped=True)
    assert np.array_equal(warped, att)                                          # identity homography
    # identity LUT on k/255 pixels is lossless, so the pipeline reduces to the 244->32 resize
    q = (np.floor(att * 255.0) / 255.0).astype(np.float32)
    ref = np.clip(oracle.resize_linear(q, 32, 32), 0, 1)
    assert np.abs(out - ref).max() <= 1.0 / 255 + 1e-6
    # integer translation = shifted copy with replicated border
    xf[0]["M"] = np.array([[1, 0, 5], [0, 1, -3], [0, 0, 1]], np.float64).reshape(9)
    _, w2 = oracle.warp_one(att, xf[0], normalize=False, want_warped=True)
    assert np.array_equal(w2[:, 0:241, 5:244], att[:, 3:244, 0:239])
    assert np.array_equal(w2[:, 243, 10], att[:, 243, 5])


def test_oracle_batch_shapes_and_ranges(oracle):
    case = gtsrb_case()
    masks, thetas = gtsrb_candidates(case, 2)
    out = oracle.warp_batch(case["vic"], masks, thetas, case["tab"][:12])
    assert out.shape == (2, 12, 3, 32, 32) and out.min() >= -0.5 and out.max() <= 0.5
    c = cifar_case()
    out = oracle.warp_batch(c["vic"], c["hull"][None], ((c["tar_s"] - c["vic_s"]) * c["hull"])[None], c["tab"][:12],
                            normalize=False)
    assert out.min() >= 0 and out.max() <= 1.0


def test_oracle_warp_geometry_against_torch_grid_sample(oracle):
    """Independent geometry check of the warp restatement (perspective_transform.py:47): the same homography sampled by
    torch's grid_sample (float64, pixel-index coordinates = align_corners=True, border padding = BORDER_REPLICATE)
    without OpenCV's 1/32-px coordinate rounding.  cvRound to 1/32 px displaces a tap by <= 1/64 px per axis, which moves
    a bilinear sample by at most (1/64 + 1/64) x the largest step between neighbouring pixels around the tap."""
    import torch
    import torch.nn.functional as F
    for case, tsel in ((gtsrb_case(), range(0, 100, 3)), (cifar_case(), range(0, 100, 9))):
        att = case["vic"]
        _, H, W = att.shape
        img = torch.from_numpy(att).double()[None]
        # largest difference to any 8-neighbour, dilated once more so that it covers all four taps of a sample
        pad = F.pad(img, (1, 1, 1, 1), mode="replicate")
        step = torch.zeros_like(img)
        for dy in range(3):
            for dx in range(3):
                step = torch.maximum(step, (pad[:, :, dy:dy + H, dx:dx + W] - img).abs())
        step = F.max_pool2d(step, 3, 1, 1)
        worst = 0.0
        for t in tsel:
            xf = case["tab"][t]
            cx, cy = int(xf["crop_x"]), int(xf["crop_y"])
            _, warped = oracle.warp_one(att, xf, normalize=case["pt_file"] is not None, want_warped=True)
            Mi = np.linalg.inv(xf["M"].reshape(3, 3))
            ys, xs = np.mgrid[0:cy, 0:cx].astype(np.float64)
            den = Mi[2, 0] * xs + Mi[2, 1] * ys + Mi[2, 2]
            sx = (Mi[0, 0] * xs + Mi[0, 1] * ys + Mi[0, 2]) / den
            sy = (Mi[1, 0] * xs + Mi[1, 1] * ys + Mi[1, 2]) / den
            grid = torch.from_numpy(np.stack([2 * sx / (W - 1) - 1, 2 * sy / (H - 1) - 1], -1))[None]
            ref = F.grid_sample(img, grid, mode="bilinear", padding_mode="border", align_corners=True)[0].numpy()
            near = F.grid_sample(step, grid, mode="nearest", padding_mode="border", align_corners=True)[0].numpy()
            err = np.abs(warped - ref)
            assert (err <= near / 32.0 + 2e-6).all(), (t, float((err - near / 32.0).max()))
            worst = max(worst, float(err.max()))
        assert worst > 1e-5              # the two samplers are NOT the same code: the quantisation is visible ...
        assert worst < 0.1               # ... and small
