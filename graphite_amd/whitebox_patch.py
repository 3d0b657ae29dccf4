"""Whitebox square-patch attack (reference whitebox/whitebox_patch.py): for each candidate position of a
mask_size x mask_size patch, `iters` EOT sign-gradient steps inside the patch, keeping the iterate with the highest
transform robustness; the best position wins.  Every gradient and every evaluation runs on the GPU
(engine.eot_gradient = gr_wb_forward + gr_model_input_grad + gr_wb_backward; eval() = gr_wb_forward + gr_model_forward).

The reference is a script (argparse + module-level loop, :39-192); this module keeps its parameters, defaults and control
flow as a function:
  * positions (:124-137): `limited_loc` (default) = the four corners and the centre; otherwise the script means "every
    8th (244 frame) / every (32 frame) offset" but builds those lists from undefined names and raises -- here that case
    walks the diagonal (the script zips its x and y lists), stride 8 or 1;
  * `whole_attack_32`: frame, mask and patch live at 32 x 32 (:99-101, :140-141);
  * per iteration (:155-183): avg gradient over the transforms, NaN -> 0, optional shrink to 32 x 32 and back around the
    sign (`gradient_shrink`, 244 frame only), step, clamp, eval over the same transforms;
  * across positions (:185-190): a position replaces the best one when its best iterate's TR is higher.
Not carried over: PNG output (the caller gets tensors).
"""
import torch

from . import _hip, engine
from .evaluator import TREvaluator
from .transforms import build_wb_table, get_transform_params
from .whitebox_attack import wb_transform_robustness


def patch_positions(frame, mask_length, limited_loc=True, whole_attack_32=False):
    """(y, x) top-left corners in the script's order (:124-137)."""
    if limited_loc:
        c = frame // 2 - int(round(mask_length / 2))
        e = frame - mask_length
        xs = [0, 0, e, e, c]
        ys = [0, e, 0, e, c]
        return list(zip(ys, xs))
    step = 1 if whole_attack_32 else 8
    return [(i, i) for i in range(0, frame - mask_length, step)]


def whitebox_patch_attack(model, img, target_label, pt_file, mask_size, num_xforms=100, step_size=0.0156862745, iters=200,
                          limited_loc=True, gradient_shrink=False, whole_attack_32=False, xforms=None, model_type='GTSRB',
                          verbose=False):
    """img [3,H,H] in [0,1] (the 244 x 244 victim; with whole_attack_32 its 32 x 32 version).  Returns a dict: the best
    position's best iterate (`adversarial`), its mask, transform robustness and position, per-position results, and the
    number of classifier queries (2 T per iteration: gradient + evaluation)."""
    normalize = model_type == 'GTSRB'                                        # :161-162
    img = _hip.to_dev(img)
    dev = img.device
    frame = img.shape[1]
    if xforms is None:
        xforms = get_transform_params(num_xforms, model_type)                 # :104
    tab = build_wb_table(xforms, img.shape[2], img.shape[1], pt_file)
    T = len(tab)
    best = dict(transform_robustness=-1, adversarial=img.clone(), mask=None, position=None)
    per_position, queries = [], 0
    for (i, j) in patch_positions(frame, mask_size, limited_loc, whole_attack_32):
        mask = torch.zeros((3, frame, frame), device=dev)
        mask[:, i:i + mask_size, j:j + mask_size] = 1
        adv = torch.clamp(img.clone(), 0, 1)
        max_tr, best_img = -1, None
        for _ in range(iters):
            avg_grad, _ = engine.eot_gradient(model, img, adv, mask, tab, target_label, normalize=normalize)
            g = torch.nan_to_num(avg_grad, nan=0.0, posinf=float("inf"), neginf=float("-inf"))   # only NaN is zeroed (:166)
            if gradient_shrink and not whole_attack_32:
                small = TREvaluator.resize(g.cpu(), 32, 32)
                g = TREvaluator.resize(torch.sign(small), frame, frame).to(dev)
            else:
                g = torch.sign(g)
            adv = torch.clamp(adv - mask * step_size * g, 0, 1)               # :175-176
            tr = wb_transform_robustness(model, img, adv, mask, tab, target_label, normalize)
            queries += 2 * T
            if tr > max_tr:
                max_tr, best_img = tr, adv.clone()
            if verbose:
                print("position (%d, %d) | Transform_Robustness: %.2f%%" % (i, j, tr * 100))
        per_position.append(dict(position=(i, j), transform_robustness=max_tr))
        if max_tr > best["transform_robustness"]:                             # :185-190
            best = dict(transform_robustness=max_tr, adversarial=best_img, mask=mask, position=(i, j))
    best.update(per_position=per_position, queries=queries)
    return best
