"""Seeded inputs shared by the CPU and GPU tests (SURVEY 8d configurations, synthetic weights).  The scenes themselves
live in the package (graphite_amd/scenes.py); here the 244 -> 32 resize is the oracle's, so CPU tests need no GPU."""
import functools
import os

from graphite_amd import scenes
from graphite_amd.scenes import INP, PT14                     # noqa: F401  (re-exported for the tests)

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _resize_np(x, dh, dw):
    import oracle_lib
    return oracle_lib.resize_linear(x, dh, dw)


@functools.lru_cache(maxsize=None)
def gtsrb_case(target_id=1, num_xforms=100, seed=42, baseline=False):
    """Config 1: stop sign (14) -> `target_id`, hull mask, seed-42 transforms."""
    return scenes.stop_sign_scene(num_xforms, target_id, seed, baseline, resize=_resize_np)


def gtsrb_candidates(case, n, tag="cand"):
    """hull itself + n-1 random 4x4-patch sub-masks, theta = (target - victim_small) * mask."""
    return scenes.stop_sign_candidates(case, n, tag)


cifar_case = scenes.cifar_scene
