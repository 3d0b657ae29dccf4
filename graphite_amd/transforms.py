"""Transform sampling and transformed-image generation -- drop-in for the
reference's `transforms.py` (black-box functions) with the pixel work on the GPU.

Reference interface kept (same names, positional order, defaults, return types):
  dist2pixels            transforms.py:13
  convert2Network        transforms.py:18   (net_size == 32 branch; ALPR branch out of scope)
  apply_transformation   transforms.py:50
  get_transform_params   transforms.py:80   (+ `patchguard=` selects pg_transforms.py:67-94)
  add_noise              transforms.py:143
  get_transformed_images transforms.py:162
  transform_wb           transforms.py:176  (differentiable twin; HIP forward + backward behind torch.autograd)
  convert2NetworkWB      transforms.py:209  (net_size == 32 branch)
Added for the batched path: `build_xform_table`, `gamma_table`, `build_wb_table`, `transform_wb_batch`.

Host side = parameter sampling (numpy's global legacy RNG, draw order identical
to the reference so seeded runs pick the same transforms) and the per-transform
descriptor table.  Everything that touches pixels goes through `_hip` (the
C-ABI HIP library); there is no CPU fallback.
"""
import math

import numpy as np
import torch

from .perspective_transform import perspective_matrix

# Must match `gr_xform` in include/graphite_hip.h (and `or_xform` in the oracle).
XFORM_DTYPE = np.dtype([("M", np.float64, (9,)), ("crop_x", np.int32), ("crop_y", np.int32),
                        ("blur", np.int32), ("pad_", np.int32), ("lut", np.uint8, (256,))], align=True)
assert XFORM_DTYPE.itemsize == 344


def dist2pixels(dist, width, obj_width=30):
    """Feet -> pixels for an object `obj_width` inches wide spanning `width` pixels."""
    return 1.0 * (dist * 12) * width / obj_width


# (angle range, baseline angle range, dist range, baseline dist max, gamma max, blur kernels,
#  baseline blur kernels, obj_width, focal); one row per parameter table of the reference.
_TABLES = {
    "GTSRB": dict(angle=50, angle_base=40, dist=(3.0, 15.0), dist_base=3.0, gamma_hi=3.5,
                  blurs=(0, 3, 5, 7), blurs_base=(0, 3)),
    "CIFAR": dict(angle=30, angle_base=None, dist=(3.0, 3.0), dist_base=3.0, gamma_hi=2.0,
                  blurs=(0,), blurs_base=(0,)),
    "PATCHGUARD": dict(angle=50, angle_base=10, dist=(3.0, 15.0), dist_base=3.0, gamma_hi=1.0,
                       blurs=(0, 3, 5, 7), blurs_base=(0, 3)),
}


def get_transform_params(num_xforms, model_type='GTSRB', nps=False, baseline=False, patchguard=False):
    """Sample `num_xforms` tuples (angle, dist, gamma, blur, crop_percent, crop_off_x,
    crop_off_y, obj_width, focal, nps) from `np.random`'s global stream.

    Draw order per transform (GTSRB/CIFAR/PatchGuard): angle[, baseline angle], dist,
    gamma, flip, crop_percent, crop_off_x, crop_off_y, blur.  Draws that cannot change
    the value (CIFAR's U(3,3) distance, its one-entry blur list) are still consumed,
    as they are in the reference.  Any other model_type uses the ALPR table (:123-139).
    """
    u = np.random.uniform
    out = []
    key = "PATCHGUARD" if patchguard else model_type
    if key in _TABLES:
        tb = _TABLES[key]
        if patchguard and model_type != 'CIFAR':
            raise AssertionError("pg_transforms supports model_type='CIFAR' only")
        blurs = tb["blurs_base"] if baseline else tb["blurs"]
        for _ in range(num_xforms):
            angle = u(-tb["angle"], tb["angle"])
            if baseline and tb["angle_base"] is not None:
                angle = u(-tb["angle_base"], tb["angle_base"])
            dist = u(tb["dist"][0], tb["dist_base"] if baseline else tb["dist"][1])
            gamma = u(1.0, tb["gamma_hi"])
            if int(round(u(0.0, 1.0))) == 1:
                gamma = 1.0 / gamma
            crop_percent = u(-0.03125, 0.03125)
            crop_off_x = u(-0.03125, 0.03125)
            crop_off_y = u(-0.03125, 0.03125)
            blur = blurs[int(math.floor(u(0.0, 1.0) * len(blurs)))]
            out.append((angle, dist, gamma, blur, crop_percent, crop_off_x, crop_off_y, 30, 3, nps))
        return out
    for _ in range(num_xforms):                     # license-plate table: different draw order
        angle = u(-15, 15)
        crop_percent = u(-0.03125, 0.03125)
        crop_off_x = u(-0.03125, 0.03125)
        crop_off_y = u(-0.03125, 0.03125)
        gamma = u(1.0, 3.5)
        if int(round(u(0.0, 1.0))) == 1:
            gamma = 1.0 / gamma
        blur = (0, 3, 5)[int(math.floor(u(0.0, 1.0) * 3))]
        dist = u(10.0, 15.0)
        out.append((angle, dist, gamma, blur, crop_percent, crop_off_x, crop_off_y, 120, 10, nps))
    return out


def normalizes(model_type):
    """Does convert2Network subtract 0.5 for this model family?  GTSRB does (transforms.py:27-29), CIFAR does not;
    'PATCHGUARD' names the pg_transforms.py twin of the CIFAR path, which always does (pg_transforms.py:20-34)."""
    return model_type in ('GTSRB', 'PATCHGUARD')


def draw_transforms(num_xforms, model_type):
    """The transform list a search driver draws for itself: get_transform_params(num_xforms, model_type)
    (generate_mask.py:41, boost.py:23); the PatchGuard drivers draw the baseline pg table instead
    (pg_generate_mask.py:43, pg_boost.py:45)."""
    if model_type == 'PATCHGUARD':
        return get_transform_params(num_xforms, 'CIFAR', baseline=True, patchguard=True)
    return get_transform_params(num_xforms, model_type)


def gamma_table(gamma):
    """256-entry uint8 lighting table: trunc(clip((i/255)**gamma * 255, 0, 255))  (transforms.py:70-72)."""
    # The reference evaluates Python's pow (libm) per entry and truncates.  numpy's vectorised power may differ from libm
    # in the last bits, which matters only where the value sits on an integer boundary: those entries (within 1e-6 of an
    # integer -- ten orders of magnitude above any pow rounding difference) are recomputed with Python's pow.
    v = np.clip(np.power(_LEVELS, float(gamma)) * 255.0, 0.0, 255.0)
    near = np.nonzero(np.abs(v - np.rint(v)) < 1e-6)[0]
    t = v.astype(np.uint8)
    for i in near:
        t[i] = int(min(max(pow(i / 255.0, gamma) * 255.0, 0.0), 255.0))
    return t


_LEVELS = np.arange(256, dtype=np.float64) / 255.0


def _gamma_table_scalar(gamma):
    """The reference's loop, entry by entry (transforms.py:70-72); tests compare gamma_table against it."""
    t = np.empty(256, np.uint8)
    for i in range(256):
        t[i] = int(min(max(pow(i / 255.0, gamma) * 255.0, 0.0), 255.0))
    return t


def build_xform_table(transforms, w, h, pt_file):
    """Per-transform descriptors for the batched kernels: forward homography, crop, blur, gamma LUT.

    Equivalent of what apply_transformation (:64-72) and get_perspective_transform (:11-44)
    compute per call, done once per transform list instead of once per (candidate, transform).
    """
    tab = np.zeros(len(transforms), dtype=XFORM_DTYPE)
    for i, tr in enumerate(transforms):
        angle, dist, gamma, blur, crop_percent, crop_off_x, crop_off_y, obj_width, focal, nps = tr
        if nps:
            raise AssertionError("nps transforms are not supported (reference asserts too, transforms.py:52-53)")
        d = dist2pixels(dist, w, obj_width)
        f = dist2pixels(focal, w, obj_width)
        M, cx, cy = perspective_matrix(angle, w, h, f, d, crop_percent, crop_off_x, crop_off_y, pt_file)
        tab[i]["M"] = M.reshape(9)
        tab[i]["crop_x"], tab[i]["crop_y"], tab[i]["blur"] = cx, cy, int(blur)
        tab[i]["lut"] = gamma_table(gamma)
    return tab


def _chw(t):
    return np.ascontiguousarray(t.detach().cpu().numpy(), dtype=np.float32)


def convert2Network(img, is_torch=True, net_size=32, model_type='GTSRB'):
    """net_size == 32 (:19-29): resize to 32x32 (bilinear, no anti-alias), clamp to [0,1], GTSRB: subtract 0.5.
    net_size != 32 (:31-45, the licence-plate branch): the short side becomes net_size, the aspect ratio is kept
    (long side int(round(net_size / short * long))), clamp only.
    `img` is a CHW tensor (is_torch) or an HWC numpy array; returns a CHW CPU tensor."""
    from . import _hip
    x = _chw(img) if is_torch else np.ascontiguousarray(np.transpose(np.asarray(img, np.float32), (2, 0, 1)))
    y = _hip.convert_to_network(x, net_size, normalizes(model_type) and net_size == 32)
    out = torch.from_numpy(y)
    assert torch.max(out) <= (0.5 if (normalizes(model_type) and net_size == 32) else 1.0)
    return out


def add_noise(image, mask, lbd, theta, return_pert_and_mask=False, clip=True):
    """comb = clip(image + lbd * upsample(theta)); with `return_pert_and_mask` also
    pert = comb inside the upsampled mask else 0, and the 0/1 upsampled mask (float64)."""
    from . import _hip
    comb, pert, mask_lg = _hip.add_noise(_chw(image), _chw(mask), _chw(theta), float(lbd), bool(clip))
    if return_pert_and_mask:
        return torch.from_numpy(comb), torch.from_numpy(pert), torch.from_numpy(mask_lg.astype(np.float64))
    return torch.from_numpy(comb)


def apply_transformation(img, mask, pert, angle, dist, gamma, blur, crop_percent, crop_off_x, crop_off_y,
                         pt_file, net_size=32, obj_width=30, focal=3, nps=False, model_type='GTSRB'):
    """One transform of an already-composited (img, mask_lg, pert) triple -> CHW tensor: [3,32,32] for net_size 32,
    [3,ny,nx] (short side net_size, clamp only) otherwise."""
    from . import _hip
    img_np, mask_np, pert_np = _chw(img), _chw(mask), _chw(pert)
    tab = build_xform_table([(angle, dist, gamma, blur, crop_percent, crop_off_x, crop_off_y, obj_width, focal, nps)],
                            img_np.shape[2], img_np.shape[1], pt_file)
    if net_size != 32:
        return torch.from_numpy(_hip.transform_composited_var(img_np, mask_np, pert_np, tab, net_size)[0])
    y = _hip.transform_composited(img_np, mask_np, pert_np, tab, net_size, normalizes(model_type))
    return torch.from_numpy(y[0])


def get_transformed_images(image, mask, transforms, lbd, theta, pt_file='inputs/GTSRB/Points/14.csv',
                           net_size=32, model_type='GTSRB'):
    """List of len(transforms) CHW CPU tensors (one with no transforms), as the reference returns: [3,32,32] for
    net_size 32; for any other net_size [3,ny,nx] with the short side = net_size (frames of any aspect ratio).

    All transforms of the candidate are produced by one fused launch; callers that only
    need hit counts should use `graphite_amd.tr_hits`, which never leaves the GPU.
    """
    from . import _hip
    img_np = _chw(image)
    if len(transforms) == 0:
        att = add_noise(image, mask, lbd, theta, True)[0]
        return [convert2Network(att, net_size=net_size, model_type=model_type)]
    tab = build_xform_table(transforms, img_np.shape[2], img_np.shape[1], pt_file)
    if net_size != 32:          # :31-45: every transform has its own [3, ny, nx] (aspect of ITS crop), clamp only
        return [torch.from_numpy(a) for a in _hip.warp_batch_var(img_np, _chw(mask)[None], _chw(theta)[None], tab,
                                                                 float(lbd), net_size)]
    out = _hip.warp_batch(img_np, _chw(mask)[None], _chw(theta)[None], tab, float(lbd), normalizes(model_type),
                          net_size)
    t = torch.from_numpy(out[0])
    return [t[i] for i in range(t.shape[0])]


# ------------------------------------------------------------------------------------------------
# differentiable ("whitebox") twin: transform_wb / convert2NetworkWB (reference :176-228)
# ------------------------------------------------------------------------------------------------
def _normal_transform_pixel(h, w):
    """kornia.geometry.conversions.normal_transform_pixel (0.5.10): pixel -> [-1, 1] coordinates, float32."""
    tr = torch.tensor([[1.0, 0.0, -1.0], [0.0, 1.0, -1.0], [0.0, 0.0, 1.0]], dtype=torch.float32)
    tr[0, 0] = tr[0, 0] * 2.0 / (w - 1 if w > 1 else 1e-14)
    tr[1, 1] = tr[1, 1] * 2.0 / (h - 1 if h > 1 else 1e-14)
    return tr


def build_wb_table(transforms, w, h, pt_file):
    """Per-transform descriptors of the whitebox kernels (gr_wb_xform): what transform_wb (:177-204) and
    kornia.warp_perspective's normalize_homography compute per call -- the float32 matrix that takes normalised
    dst coordinates to normalised src coordinates, the warp's dsize, gamma and the blur kernel size."""
    from ._hip import WB_XFORM_DTYPE
    tab = np.zeros(len(transforms), dtype=WB_XFORM_DTYPE)
    n_src_inv = torch.inverse(_normal_transform_pixel(h, w))
    for i, tr in enumerate(transforms):
        angle, dist, gamma, blur, crop_percent, crop_off_x, crop_off_y, obj_width, focal, _ = tr
        d = dist2pixels(dist, h, obj_width)            # the reference passes att.size()[2] = H here (:200-201)
        f = dist2pixels(focal, h, obj_width)
        M, cx, cy = perspective_matrix(angle, w, h, f, d, crop_percent, crop_off_x, crop_off_y, pt_file)
        M32 = torch.from_numpy(np.asarray(M, np.float64)).float()
        dst_norm_trans_src_norm = _normal_transform_pixel(cy, cx) @ (M32 @ n_src_inv)
        tab[i]["A"] = torch.inverse(dst_norm_trans_src_norm).numpy().reshape(9)
        tab[i]["crop_x"], tab[i]["crop_y"], tab[i]["gamma"], tab[i]["blur"] = cx, cy, gamma, int(blur)
    return tab


class _TransformWB(torch.autograd.Function):
    """out[T,3,net,net] = transform_wb(orig, att, mask, transform_t); differentiable w.r.t. att (HIP backward kernels)."""

    @staticmethod
    def forward(ctx, att, orig, mask, tab, net):
        from . import _hip
        ctx.save_for_backward(att, orig, mask)
        ctx.tab, ctx.net = tab, net
        return _hip.wb_forward(orig, att, mask, tab, net)

    @staticmethod
    def backward(ctx, grad_out):
        from . import _hip
        att, orig, mask = ctx.saved_tensors
        g = _hip.wb_backward(orig, att, mask, ctx.tab, grad_out.contiguous(), ctx.net)
        return g.to(att.device), None, None, None, None


def _chw3(t):
    t = torch.as_tensor(t, dtype=torch.float32)
    return t[0] if t.dim() == 4 else t


def transform_wb_batch(orig, att, mask, transforms, pt_file='inputs/GTSRB/Points/14.csv', net_size=32):
    """All transforms of one (orig, att, mask) in one launch: [T,3,32,32] on the GPU, values in [0,1];
    gradients flow to `att` (one backward launch for all T).  `transforms` is the list from get_transform_params
    or a table from build_wb_table."""
    if net_size != 32:
        raise NotImplementedError("only the net_size == 32 branch of convert2NetworkWB is implemented")
    from . import _hip
    o, a, m = _chw3(orig), _chw3(att), _chw3(mask)
    if isinstance(transforms, np.ndarray) and transforms.dtype == _hip.WB_XFORM_DTYPE:
        tab = transforms
    else:
        tab = build_wb_table(transforms, o.shape[2], o.shape[1], pt_file)
    if (tab["crop_x"] != tab["crop_y"]).any():
        raise NotImplementedError("non-square crops (ALPR) are outside this path")
    dev = _hip.to_dev(o).device
    return _TransformWB.apply(a.to(dev), o.to(dev), m.to(dev).float(), tab, net_size)


def transform_wb(orig, att, mask, transform, pt_file='inputs/GTSRB/Points/14.csv', net_size=32):
    """Reference signature (:176): orig, att [1,3,H,W]; mask [3,H,W]; one transform tuple -> [1,3,32,32] in [0,1]."""
    return transform_wb_batch(orig, att, mask, [transform], pt_file, net_size)


def convert2NetworkWB(img, is_torch=True, net_size=32):
    """kornia resize(img, 32, align_corners=False) + clamp (:209-214) for a [B,3,H,W] or [3,H,W] tensor.
    Stand-alone use only (the baselines); inside transform_wb the resize is fused into the HIP kernel.  This helper
    is plain torch (F.interpolate), on whatever device `img` lives."""
    if net_size != 32:
        raise NotImplementedError("only the net_size == 32 branch of convert2NetworkWB is implemented")
    x = img if img.dim() == 4 else img[None]
    y = torch.clamp(torch.nn.functional.interpolate(x, size=(32, 32), mode='bilinear', align_corners=False), 0.0, 1.0)
    return y if img.dim() == 4 else y[0]
