"""Seeded inputs shared by the CPU and GPU tests (SURVEY 8d configurations, synthetic weights)."""
import functools
import os

import numpy as np

from graphite_amd import synth, transforms as T
from graphite_amd.imageio import load_image_chw

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
INP = os.path.join(GOLD, "inputs")
PT14 = os.path.join(INP, "GTSRB/Points/14.csv")


def _resize_np(x, dh, dw):
    """Plain numpy bilinear (half-pixel centres) used only to build small test inputs."""
    import oracle_lib
    return oracle_lib.resize_linear(x, dh, dw)


@functools.lru_cache(maxsize=None)
def gtsrb_case(target_id=1, num_xforms=100, seed=42, baseline=False):
    """Config 1: stop sign (14) -> `target_id`, hull mask, seed-42 transforms."""
    vic = load_image_chw(os.path.join(INP, "GTSRB/images/14.png"))
    tar = load_image_chw(os.path.join(INP, f"GTSRB/images/{target_id}.png"))
    hull = (load_image_chw(os.path.join(INP, "GTSRB/Hulls/14.png")) * 255 > 128).astype(np.float32)
    vic_s, tar_s = _resize_np(vic, 32, 32), _resize_np(tar, 32, 32)
    np.random.seed(seed)
    xf = T.get_transform_params(num_xforms, "GTSRB", baseline=baseline)
    tab = T.build_xform_table(xf, 244, 244, PT14)
    return dict(vic=vic, tar_s=tar_s, vic_s=vic_s, hull=hull, xf=xf, tab=tab, pt_file=PT14)


def gtsrb_candidates(case, n, tag="cand"):
    """hull itself + n-1 random 4x4-patch sub-masks, theta = (target - victim_small) * mask."""
    masks = case["hull"][None]
    if n > 1:
        masks = np.concatenate([masks, synth.synth_patch_masks(tag, n - 1, case["hull"])])
    thetas = (case["tar_s"] - case["vic_s"])[None] * masks
    return np.ascontiguousarray(masks), np.ascontiguousarray(thetas.astype(np.float32))


@functools.lru_cache(maxsize=None)
def cifar_case(num_xforms=100, seed=42, patchguard=False, baseline=False):
    """Config 3 / 5: synthetic 32x32 victim/target (CIFAR PNGs are a network download), all-ones mask."""
    vic = synth.synth_victim("cifar-vic", 32)
    tar = synth.synth_victim("cifar-tar", 32)
    ones = np.ones((3, 32, 32), np.float32)
    np.random.seed(seed)
    xf = T.get_transform_params(num_xforms, "CIFAR", baseline=baseline, patchguard=patchguard)
    tab = T.build_xform_table(xf, 32, 32, None)
    return dict(vic=vic, tar_s=tar, vic_s=vic, hull=ones, xf=xf, tab=tab, pt_file=None)
