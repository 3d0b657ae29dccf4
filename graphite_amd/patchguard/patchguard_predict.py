"""Place of the reference's `patchguard/patchguard_predict.py` (PatchGuard / PatchGuardNoDefense).

NOT built: both classes wrap BagNet-17 (`PatchGuard.nets.bagnet`) and `masking_defense` (`PatchGuard.utils.defense_utils`)
from a clone of inspire-group/PatchGuard plus its `bagnet17_192_cifar.pth` checkpoint (README.md:113-117) -- third-party
source and weights that are neither vendored in the reference nor obtainable here.  The attack drivers only call
`model.predict(x)` on [B,3,32,32] inputs in [-0.5, 0.5] and expect int64 labels [B]; any such object can be passed to
`main_patchguard.attack_network` (the tests use the HIP WideResNet as the stand-in).
"""
from math import ceil

RF_SIZE, RF_STRIDE, PATCH_SIZE = 17, 8, 30
WINDOW_SIZE = ceil((PATCH_SIZE + RF_SIZE - 1) / RF_STRIDE)          # patchguard_predict.py:33-36 -> 6


class _Unavailable:
    def __init__(self, *a, **k):
        raise RuntimeError(type(self).__name__ + ": needs inspire-group/PatchGuard's BagNet-17, masking_defense and the "
                           "bagnet17_192_cifar.pth checkpoint, none of which ship with GRAPHITE; pass any object with a "
                           ".predict(x) method to graphite_amd.patchguard.main_patchguard.attack_network instead")


class PatchGuard(_Unavailable):
    pass


class PatchGuardNoDefense(_Unavailable):
    pass
