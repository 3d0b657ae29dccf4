"""Drop-in for the reference's `patchguard/pg_transforms.py` (same names, positional order, defaults).

  dist2pixels            pg_transforms.py:15
  convert2Network        pg_transforms.py:20-34   net_size 32 only (the reference asserts), ALWAYS normalised (x - 0.5)
  apply_transformation   pg_transforms.py:37-64
  get_transform_params   pg_transforms.py:67-94   model_type 'CIFAR' only
  add_noise              pg_transforms.py:97-112
  get_transformed_images pg_transforms.py:115-126
  transform_wb           pg_transforms.py:129-160  as transforms.transform_wb (the PatchGuard twin clamps only, no shift)
  convert2NetworkWB      pg_transforms.py:162-173
"""
from .. import transforms as _T

dist2pixels = _T.dist2pixels
add_noise = _T.add_noise


def convert2Network(img, is_torch=True, net_size=32):
    assert net_size == 32                                   # pg_transforms.py:31-32
    return _T.convert2Network(img, is_torch, net_size, model_type='PATCHGUARD')


def apply_transformation(img, mask, pert, angle, dist, gamma, blur, crop_percent, crop_off_x, crop_off_y, pt_file,
                         net_size=32, obj_width=30, focal=3, nps=False):
    assert net_size == 32
    return _T.apply_transformation(img, mask, pert, angle, dist, gamma, blur, crop_percent, crop_off_x, crop_off_y, pt_file,
                                   net_size, obj_width, focal, nps, model_type='PATCHGUARD')


def get_transform_params(num_xforms, model_type='CIFAR', nps=False, baseline=False):
    assert model_type == 'CIFAR'                            # pg_transforms.py:92-93
    return _T.get_transform_params(num_xforms, model_type, nps, baseline, patchguard=True)


def get_transformed_images(image, mask, transforms, lbd, theta, pt_file='../inputs/GTSRB/Points/14.csv', net_size=32):
    assert net_size == 32
    return _T.get_transformed_images(image, mask, transforms, lbd, theta, pt_file, net_size, model_type='PATCHGUARD')


def transform_wb(orig, att, mask, transform, pt_file='../inputs/GTSRB/Points/14.csv', net_size=32):
    return _T.transform_wb(orig, att, mask, transform, pt_file, net_size)


def convert2NetworkWB(img, is_torch=True, net_size=32):
    return _T.convert2NetworkWB(img, is_torch, net_size)
