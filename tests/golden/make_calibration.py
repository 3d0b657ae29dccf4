#!/usr/bin/env python3
"""Centre the last-layer bias of the synthetic classifiers.

Random-init nets predict one constant class for every input, which would make
every hit-count comparison trivially true.  This script runs the CPU oracle on
a calibration set (config-1 style warped stop-sign composites for GTSRBNet,
config-3 style synthetic 32x32 composites for WRN-34-10, config-5 style composites through the PatchGuard pre-processing
for BagNet-17, whose LOCAL logits are centred per class over images and positions), and stores
`bias - mean(logits)` so that the arg-max depends on the input.  Output:
graphite_amd/data/<tag>_last_bias.npy (exact float32), loaded by graphite_amd.synth.

    cd /root/repo && python3 -B tests/golden/make_calibration.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

import oracle_lib as O                                   # noqa: E402
from graphite_amd import synth, transforms as T, weights  # noqa: E402
from graphite_amd.imageio import load_image_chw           # noqa: E402

INP = os.path.join(REPO, "graphite_amd", "data", "inputs")
OUT = os.path.join(REPO, "graphite_amd", "data")


def gtsrb():
    vic = load_image_chw(os.path.join(INP, "GTSRB/images/14.png"))
    tar = load_image_chw(os.path.join(INP, "GTSRB/images/1.png"))
    hull = (load_image_chw(os.path.join(INP, "GTSRB/Hulls/14.png")) * 255 > 128).astype(np.float32)
    vic_s, tar_s = O.resize_linear(vic, 32, 32), O.resize_linear(tar, 32, 32)
    np.random.seed(42)
    tab = T.build_xform_table(T.get_transform_params(100, "GTSRB"), 244, 244,
                              os.path.join(INP, "GTSRB/Points/14.csv"))
    masks = np.concatenate([hull[None], synth.synth_patch_masks("calib", 7, hull)])
    thetas = (tar_s - vic_s)[None] * masks
    imgs = O.warp_batch(vic, masks, thetas, tab).reshape(-1, 3, 32, 32)
    sd = synth.gtsrb_state_dict("gtsrb-v1", calibrated=False)
    lg = O.gtsrb_forward(weights.gtsrb_tensor_list(sd), imgs)
    bias = (sd["fc7_1.bias"].astype(np.float64) - lg.astype(np.float64).mean(0)).astype(np.float32)
    np.save(os.path.join(OUT, "gtsrb-v1_last_bias.npy"), bias)
    sd["fc7_1.bias"] = bias
    p = O.gtsrb_forward(weights.gtsrb_tensor_list(sd), imgs).argmax(1)
    print("gtsrb: classes predicted on the calibration set:", np.count_nonzero(np.bincount(p, minlength=43)))


def wrn():
    vic = synth.synth_victim("cifar-calib", 32)
    tar = synth.synth_victim("cifar-calib-target", 32)
    ones = np.ones((3, 32, 32), np.float32)
    np.random.seed(42)
    tab = T.build_xform_table(T.get_transform_params(24, "CIFAR"), 32, 32, None)
    masks = np.concatenate([ones[None], synth.synth_patch_masks("calib-c", 3, ones)])
    thetas = (tar - vic)[None] * masks
    imgs = O.warp_batch(vic, masks, thetas, tab, normalize=False).reshape(-1, 3, 32, 32)
    sd = synth.wrn_state_dict("wrn-v1", calibrated=False)
    lg = O.wrn_forward(weights.wrn_tensor_list(sd), 34, 10, 10, imgs)
    bias = (sd["fc.bias"].astype(np.float64) - lg.astype(np.float64).mean(0)).astype(np.float32)
    np.save(os.path.join(OUT, "wrn-v1_last_bias.npy"), bias)
    sd["fc.bias"] = bias
    p = O.wrn_forward(weights.wrn_tensor_list(sd), 34, 10, 10, imgs).argmax(1)
    print("wrn: class histogram on the calibration set:", np.bincount(p, minlength=10))


def bagnet():
    vic = synth.synth_victim("cifar-calib", 32)
    tar = synth.synth_victim("cifar-calib-target", 32)
    ones = np.ones((3, 32, 32), np.float32)
    np.random.seed(42)
    tab = T.build_xform_table(T.get_transform_params(12, "CIFAR", baseline=True, patchguard=True), 32, 32, None)
    masks = np.concatenate([ones[None], synth.synth_patch_masks("calib-pg", 3, ones)])
    thetas = (tar - vic)[None] * masks
    imgs = O.warp_batch(vic, masks, thetas, tab, normalize=True).reshape(-1, 3, 32, 32)
    sd = synth.bagnet_state_dict("bagnet-v1", calibrated=False)
    lo = O.bagnet_forward(weights.bagnet_tensor_list(sd), O.pg_preprocess(imgs))
    bias = (sd["fc.bias"].astype(np.float64) - lo.astype(np.float64).mean((0, 1, 2))).astype(np.float32)
    np.save(os.path.join(OUT, "bagnet-v1_last_bias.npy"), bias)
    sd["fc.bias"] = bias
    lo = O.bagnet_forward(weights.bagnet_tensor_list(sd), O.pg_preprocess(imgs))
    print("bagnet17: robust (masking_defense) class histogram on the calibration set:",
          np.bincount(O.masking_defense(lo)[1], minlength=10), "undefended:", np.bincount(O.pg_mean_logits(lo).argmax(1), minlength=10))


if __name__ == "__main__":
    which = sys.argv[1:] or ["gtsrb", "wrn", "bagnet"]
    for name in which:
        {"gtsrb": gtsrb, "wrn": wrn, "bagnet": bagnet}[name]()
