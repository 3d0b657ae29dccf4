"""The scenes of SURVEY 8d as the package ships them: the reference's own stop-sign inputs (inputs/GTSRB/{images,Hulls,
Points}, cifar/masks/mask.png -- data files, kept under graphite_amd/data/inputs) and the seeded synthetic 32x32 CIFAR
pair (the CIFAR PNGs are a network download, cifar/create_cifar_vts.py:10).  bench.py, the scripts and the tests build
their workloads from here."""
import functools
import os

import numpy as np

from . import synth, transforms as T
from .imageio import load_image_chw

INP = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "inputs")
PT14 = os.path.join(INP, "GTSRB/Points/14.csv")


def gtsrb_image(label):
    return os.path.join(INP, "GTSRB/images/%s.png" % label)


HULL14 = os.path.join(INP, "GTSRB/Hulls/14.png")
CIFAR_MASK = os.path.join(INP, "cifar/masks/mask.png")


def load_hull(path=HULL14):
    """main.py:54: mask = imread(...) > 128 (float 0/1, three identical channels)."""
    return (load_image_chw(path) * 255 > 128).astype(np.float32)


def stop_sign_scene(num_xforms=100, target_id=1, seed=42, baseline=False, resize=None):
    """Config 1 / 4: stop sign (14) -> `target_id`, hull mask 14, Points/14.csv, seed-`seed` GTSRB transforms.
    `resize` = the 244 -> 32 float resize to use (default: the HIP library's, so that no checker code is involved)."""
    if resize is None:
        from . import _hip
        resize = _hip.resize_linear
    vic = load_image_chw(gtsrb_image(14))
    tar = load_image_chw(gtsrb_image(target_id))
    np.random.seed(seed)
    xf = T.get_transform_params(num_xforms, "GTSRB", baseline=baseline)
    return dict(vic=vic, tar_s=resize(tar, 32, 32), vic_s=resize(vic, 32, 32), hull=load_hull(), xf=xf,
                tab=T.build_xform_table(xf, 244, 244, PT14), pt_file=PT14)


@functools.lru_cache(maxsize=None)
def cifar_scene(num_xforms=100, seed=42, patchguard=False, baseline=False):
    """Config 3 / 5: synthetic 32x32 victim / target, all-ones mask, no outline file."""
    vic = synth.synth_victim("cifar-vic", 32)
    tar = synth.synth_victim("cifar-tar", 32)
    np.random.seed(seed)
    xf = T.get_transform_params(num_xforms, "CIFAR", baseline=baseline, patchguard=patchguard)
    return dict(vic=vic, tar_s=tar, vic_s=vic, hull=np.ones((3, 32, 32), np.float32), xf=xf,
                tab=T.build_xform_table(xf, 32, 32, None), pt_file=None)


def stop_sign_candidates(scene, n, tag="cand", include_hull=True):
    """The hull itself (optional) + random 4x4-patch sub-masks of it; theta = (target - victim_small) * mask."""
    if include_hull:
        masks = scene["hull"][None]
        if n > 1:
            masks = np.concatenate([masks, synth.synth_patch_masks(tag, n - 1, scene["hull"])])
    else:
        masks = synth.synth_patch_masks(tag, n, scene["hull"])
    thetas = (scene["tar_s"] - scene["vic_s"])[None] * masks
    return np.ascontiguousarray(masks), np.ascontiguousarray(thetas.astype(np.float32))
