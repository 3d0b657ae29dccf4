"""Coarse mask reduction -- drop-in for the reference's `coarse_reduction.py`.

  get_coarse_reduced_mask           coarse_reduction.py:12-72   (same signature, same 7-tuple)
  perform_linear_coarse_reduction   coarse_reduction.py:75-104
  perform_binary_coarse_reduction   coarse_reduction.py:107-165

Decisions and query counts are the reference's: the leftmost-pivot binary search (including its
final re-evaluation of the pivot, :163) is replayed evaluation by evaluation; each evaluation is one
fused GPU pass over the transforms.  A 100-image launch is latency-bound on an MI355X, so with the fused
evaluator the pivots of the next three levels (7 candidates, every outcome) share a launch and the search
is replayed on the cached results; a duck-typed classifier is queried pivot by pivot as in the reference.
"""
import time

import torch

from .evaluator import TREvaluator, make_evaluator


def _union(best_mask, patch):
    nm = best_mask + (torch.zeros(best_mask.size()) + patch)
    return torch.where(nm > 0, torch.ones(best_mask.size()), torch.zeros(best_mask.size()))


def _theta(init_theta, img_t_small, img_v_small, mask):
    return (img_t_small - img_v_small) * mask if init_theta is None else init_theta * mask


def get_coarse_reduced_mask(start_mask, object_size, patches, indices, img_v, img_t, lbl_v, lbl_t, model, xforms, pt_file,
                            net_size, num_xforms=1000, patch_size=4, err_threshold=0.5, coarse_red_mode='binary',
                            direction="forward", args=None, debug=False, model_type='GTSRB', init_theta=None,
                            _evaluator=None):
    if args is not None:
        coarse_red_mode = args.coarse_mode
    ev = make_evaluator(_evaluator, model, img_v, xforms, pt_file, model_type, lbl_v, lbl_t)
    q = 0
    img_v_small = ev.resize(img_v, start_mask.size()[1], start_mask.size()[2])
    if img_t.size()[1] != img_v_small.size()[1] or img_t.size()[2] != img_v_small.size()[2]:
        img_t_small = ev.resize(img_t, start_mask.size()[1], start_mask.size()[2])
    else:
        img_t_small = img_t

    success_rate, query_ct = ev.error(start_mask, _theta(init_theta, img_t_small, img_v_small, start_mask))
    q += query_ct
    print("init_tr", 1 - success_rate)
    print("init bits", start_mask.sum() / 3)

    if "binary" not in coarse_red_mode:
        pivot, best_mask, best_tr, query_ct = perform_linear_coarse_reduction(
            patches, err_threshold, lbl_v, lbl_t, img_t_small, img_v_small, img_v, model, xforms, start_mask, pt_file,
            net_size, model_type=model_type, init_theta=init_theta, _evaluator=ev)
    else:
        pivot, best_mask, best_tr, query_ct = perform_binary_coarse_reduction(
            patches, err_threshold, lbl_v, lbl_t, img_t_small, img_v_small, img_v, model, xforms, start_mask, pt_file,
            net_size, direction, model_type=model_type, init_theta=init_theta, _evaluator=ev)
    patches, indices = patches[:], indices[:]
    q += query_ct
    print("COARSE_REDUCTION: mode", coarse_red_mode, "patches incorporated", pivot, "bits", (best_mask.sum() / 3).item(),
          "tr", best_tr, "queries", q)
    return 9999999, best_tr, best_mask, q, pivot, patches, indices


def perform_linear_coarse_reduction(patches, err_threshold, lbl_v, lbl_t, img_t_small, img_v_small, img_v, model, xforms,
                                    start_mask, pt_file, net_size, model_type='GTSRB', init_theta=None, _evaluator=None):
    ev = make_evaluator(_evaluator, model, img_v, xforms, pt_file, model_type, lbl_v, lbl_t)
    best_mask, best_tr, tot = torch.zeros(start_mask.size()), 0, 0
    j = 0
    for j in range(len(patches)):
        next_mask = _union(best_mask, patches[j])
        if (next_mask - best_mask).sum() == 0:
            continue
        success_rate, query_ct = ev.error(next_mask, _theta(init_theta, img_t_small, img_v_small, next_mask))
        tot += query_ct
        best_mask, best_tr = next_mask, 1 - success_rate
        if best_tr >= (1 - err_threshold):
            break
    return j, best_mask, best_tr, tot


COARSE_SPEC_DEPTH = 3     # levels of the binary search evaluated per launch (2**3 - 1 = 7 pivots) when speculation is allowed


def perform_binary_coarse_reduction(patches, err_threshold, lbl_v, lbl_t, img_t_small, img_v_small, img_v, model, xforms,
                                    start_mask, pt_file, net_size, direction="forward", model_type='GTSRB',
                                    init_theta=None, _evaluator=None):
    ev = make_evaluator(_evaluator, model, img_v, xforms, pt_file, model_type, lbl_v, lbl_t)
    tot = 0
    n = len(patches)
    # prefix unions: mask_upto(k) = union of the first k patches (the reference rebuilds it from scratch per pivot,
    # :128-136); patches are 0/1, so the union is "running sum > 0".  The running sums are built lazily from the nearest
    # smaller prefix already made and only the pivots actually evaluated are kept (O(log n) masks, not n).
    prefix = {0: torch.zeros(start_mask.size())}

    def mask_upto(pivot):
        pivot = max(0, min(int(pivot), n))
        if pivot not in prefix:
            q = max(k for k in prefix if k <= pivot)
            acc = prefix[q].clone()
            for j in range(q, pivot):
                acc += patches[j]
            prefix[pivot] = acc
        return torch.where(prefix[pivot] > 0, torch.ones(start_mask.size()), torch.zeros(start_mask.size()))

    # A launch of ONE 100-image candidate leaves the GPU mostly idle, so with an evaluator that may speculate the pivots
    # of the next COARSE_SPEC_DEPTH levels (every outcome) are evaluated together and the search is replayed on the
    # cached results.  Only pivots the reference visits are counted (tot), its final re-evaluation included (:163).
    spec = getattr(ev, "speculate", False) and hasattr(ev, "errors")
    cache = {}

    def subtree(lo, hi, depth):
        if lo > hi or depth == 0:
            return []
        mi = lo + (hi - lo) // 2
        out = [mi]
        if mi > 0:
            out += subtree(lo, mi - 1, depth - 1)
        return out + subtree(mi + 1, hi, depth - 1)

    def tr_at(pivot, lo=None, hi=None):
        nonlocal tot
        if not spec:
            m = mask_upto(pivot)
            success_rate, query_ct = ev.error(m, _theta(init_theta, img_t_small, img_v_small, m))
            tot += query_ct
            return 1 - success_rate, m
        if pivot not in cache:
            want = [pivot] if lo is None else subtree(lo, hi, COARSE_SPEC_DEPTH)
            want = sorted(set(p for p in want if p not in cache) | {pivot})
            masks = [mask_upto(p) for p in want]
            res = ev.errors(masks, [_theta(init_theta, img_t_small, img_v_small, m) for m in masks])
            for p, m, (success_rate, query_ct) in zip(want, masks, res):
                cache[p] = (1 - success_rate, query_ct, m)
        tr, query_ct, m = cache[pivot]
        tot += query_ct
        return tr, m.clone()

    threshold = 1 - err_threshold
    mi = 0 if n == 1 else -1
    if mi < 0:
        lo, hi = 0, n - 1
        while lo <= hi:
            mi = lo + (hi - lo) // 2
            score, _ = tr_at(mi, lo, hi)
            if score >= threshold:
                if mi > 0:
                    hi = mi - 1
                    continue
                break
            lo, mi = mi + 1, mi + 1
    pivot = mi
    assert 0 <= pivot <= n
    best_tr, best_mask = tr_at(pivot)                      # the reference's final re-evaluation (:163)
    return pivot, best_mask, best_tr, tot
