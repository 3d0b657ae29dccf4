"""Whitebox GRAPHITE (reference whitebox/whitebox_attack.py): EOT sign-gradient attack inside a mask that is shrunk
patch by patch, every gradient and every transform-robustness evaluation on the GPU.

The reference is a script (argparse + module-level loop, :39-end); this module keeps its parameters, defaults and
control flow as a function.  Per inner iteration the reference runs, for each of the T transforms,
transform_wb -> model -> CrossEntropy -> autograd.grad (:149-160) and then T more forwards for eval() (:22-36);
here that is engine.eot_gradient (gr_wb_forward + gr_model_input_grad + gr_wb_backward) and one
gr_wb_forward + gr_model_forward.  Query accounting (`num_queries`) follows :161 and :176.

Both mask-shrinking branches are here: patch removal (:192-221, the script's hard-wired default) and the pixel-wise
branch (:183-189, :222-236; `patch_removal=False`), which drops the `pixels_per_round` pixels with the smallest signed
gradient sum per round and, when a round fails, restores the previous mask and halves that number.
`model_type='CIFAR'` runs the same loop on a WideResNet (no -0.5 shift, :154-155; input gradients from
gr_model_input_grad's WideResNet pass).
Not carried over: PNG outputs (the caller gets tensors); --gradient_shrink's cv2 round trip (:165-169) is done with the
HIP resize.
"""
import numpy as np
import torch

from . import _hip, engine
from .evaluator import TREvaluator
from .transforms import build_wb_table, get_transform_params


def wb_transform_robustness(model, img, adv, mask, tab, target, normalize=True):
    """eval() of the reference (:22-36): fraction of transforms whose arg-max is the target label."""
    x = _hip.wb_forward(img, adv, mask, tab)
    if normalize:
        x = x - 0.5
    preds = getattr(model, "hip_model", model).forward(x).argmax(dim=1)
    return float((preds == int(target)).sum().item()) / len(tab)


def _remove_patches(mask, score, size, interval, count):
    """:203-223: `count` times, zero the size x size window (on the reference's np.arange grid) whose mean score over
    its still-masked pixels is smallest."""
    h, w = score.shape
    removed = []
    for _ in range(count):
        best, best_idx = None, None
        for i in np.arange(0, h - size + 0.0001, interval):
            for j in np.arange(0, w - size + 0.0001, interval):
                i0, i1, j0, j1 = int(round(i)), int(round(i + size)), int(round(j)), int(round(j + size))
                msum = float(mask[0, i0:i1, j0:j1].sum())
                if msum > 0:
                    g = float(score[i0:i1, j0:j1].sum()) / msum
                    if best is None or g < best:
                        best, best_idx = g, (i0, i1, j0, j1)
        if best_idx is None:
            break
        i0, i1, j0, j1 = best_idx
        mask[:, i0:i1, j0:j1] = 0
        removed.append((i0, j0))
    return removed


def _remove_pixels(mask, grad, count):
    """:222-236.  `count` times: g = mask * g; s = sum over channels (signed); s == 0 or NaN -> 999999; zero the mask at
    the FIRST minimum of s.  A removed pixel turns into 999999 and nothing else changes, so the loop is a stable sort:
    the `count` smallest live values go, in order; once only 999999 entries are left flat index 0 is hit again and
    again (idempotent).  Returns the number of pixels whose mask changed."""
    s = torch.sum(mask * grad, dim=0)
    s = torch.where(s == 0, torch.full_like(s, 999999.0), s)
    s = torch.nan_to_num(s, nan=999999.0, posinf=float("inf"), neginf=float("-inf"))
    flat = s.flatten()
    order = torch.sort(flat, stable=True).indices
    live = int((flat < 999999.0).sum().item()) if bool((flat <= 999999.0).all()) else None
    if live is None:                       # a genuine value above the sentinel: follow the loop literally
        removed = 0
        for _ in range(count):
            i = int(torch.argmin(flat))
            removed += int(mask.flatten(1)[0, i] > 0)
            mask.flatten(1)[:, i] = 0
            flat[i] = 999999.0 if not bool((mask.flatten(1)[:, i] != 0).any()) else flat[i]
        return removed
    take = order[:min(count, live)]
    if count > live:
        # every entry is 999999 by now, so the loop's "first minimum" is flat index 0 from here on: pixel (0, 0) loses
        # its mask bits too (a no-op if it was live and already taken, or never in the mask)
        take = torch.cat([take, torch.zeros(1, dtype=take.dtype, device=take.device)])
    before = int((mask.flatten(1)[0, take] > 0).sum().item())
    mask.flatten(1)[:, take] = 0
    return before


def whitebox_attack(model, img, mask, target_label, pt_file, num_xforms=100, step_size=0.0156862745, iters=50, min_tr=0.8,
                    gradient_shrink=False, aligned=True, patch_removal_size=-1, patch_removal_interval=15.25,
                    first_round_iters=500, max_rounds=None, xforms=None, verbose=False, patch_removal=True,
                    pixels_per_round=5000, pixels_per_round_auto=True, model_type='GTSRB'):
    """img [3,H,W] in [0,1] (CPU or CUDA), mask [3,H,W] 0/1, model = graphite_amd.models.GTSRBNet (or WideResNet with
    model_type='CIFAR').  Returns a dict: adversarial image and mask of the last round that reached `min_tr`, its
    transform robustness, mask size (pixels), query count, rounds."""
    normalize = model_type == 'GTSRB'                                       # :154-155
    dev = _hip.to_dev(img).device
    img = _hip.to_dev(img)
    mask = _hip.to_dev(mask).clone()
    if patch_removal_size < 0:
        patch_removal_size = 30.5 if aligned else 30                        # :75-78
    if xforms is None:
        xforms = get_transform_params(num_xforms, model_type)                # :113
    tab = build_wb_table(xforms, img.shape[2], img.shape[1], pt_file)
    T = len(tab)
    num_queries = 0
    rounds = 0
    previous_mask = mask.clone()
    prev_attack = img.clone()
    prev_tr, prev_mask_size = None, None
    history = []
    while True:
        adv = torch.where(mask > 0.5, prev_attack, img)                      # :137-139
        rand_start = torch.FloatTensor(3, 32, 32).uniform_(-8 / 255, 8 / 255)
        rand_start = TREvaluator.resize(rand_start, img.shape[1], img.shape[2]).to(dev)   # cv2.resize (:141-143)
        adv = torch.clamp(adv + mask * rand_start, 0, 1)
        final_avg_grad, tr = None, 0.0
        loop_length = iters if rounds > 0 else first_round_iters           # :150
        for _ in range(loop_length):
            avg_grad, _ = engine.eot_gradient(model, img, adv, mask, tab, target_label, normalize=normalize)
            num_queries += T
            g = torch.nan_to_num(avg_grad, nan=0.0, posinf=float("inf"), neginf=float("-inf"))   # only NaN is zeroed (:164)
            if gradient_shrink:
                small = TREvaluator.resize(g.cpu(), 32, 32)
                g = TREvaluator.resize(torch.sign(small), img.shape[1], img.shape[2]).to(dev)
            else:
                g = torch.sign(g)
            adv = torch.clamp(adv - mask * step_size * g, 0, 1)              # :172-174
            tr = wb_transform_robustness(model, img, adv, mask, tab, target_label, normalize)
            num_queries += T
            final_avg_grad = avg_grad
            if verbose:
                print("Iteration | Transform_Robustness: {:.2f}%".format(tr * 100))
            if tr >= min_tr:
                break
        # :183-189: a failed pixel-wise round puts the previous mask back and halves the step (no new round is counted)
        if not patch_removal and pixels_per_round_auto and tr < min_tr:
            mask = previous_mask.clone()
            pixels_per_round //= 2
            history.append(dict(round=rounds, tr=tr, reset=True, pixels_per_round=pixels_per_round))
            if pixels_per_round == 0:                                        # the reference would spin here for ever
                break
            continue
        if patch_removal:
            # :192-221: saliency = |mask * grad * perturbation| summed over channels, four patches per round
            pert = adv - img
            fg = torch.nan_to_num(final_avg_grad, nan=0.0, posinf=float("inf"), neginf=float("-inf"))
            score = torch.sum(torch.abs(mask * fg * pert), dim=0).cpu().numpy()
            mask_c = mask.cpu()
            removed = _remove_patches(mask_c, score, patch_removal_size, patch_removal_interval, 4)
            mask = mask_c.to(dev)
        else:
            removed = _remove_pixels(mask, final_avg_grad, pixels_per_round)   # :222-236
        history.append(dict(round=rounds, tr=tr, removed=removed))
        if tr < min_tr:                                                      # :246-253: keep the previous round's result
            break
        previous_mask = mask.clone()
        prev_attack = adv.detach().clone()
        prev_tr = tr
        prev_mask_size = int(round(float((mask > 0.5).sum().item()) / 3))
        rounds += 1
        if max_rounds is not None and rounds >= max_rounds:
            break
    return dict(adversarial=prev_attack, mask=previous_mask, transform_robustness=prev_tr, mask_size=prev_mask_size,
                queries=num_queries, rounds=rounds, history=history)
