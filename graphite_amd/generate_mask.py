"""Mask generation -- drop-in for the reference's `generate_mask.py`.

  generate_mask          generate_mask.py:20-145   (same signature, same 5-tuple)
  get_heatmap            generate_mask.py:148-178
  get_fine_reduced_mask  generate_mask.py:181-250
  survey_heatmap         generate_mask.py:254-267

Stage 1 (heatmap) evaluates every "hull minus one patch" candidate (748 for the stop sign) in ONE
fused launch instead of one Python round trip each.  Stage 2 delegates to coarse_reduction.  Stage 3
(greedy fine reduction) is sequentially dependent -- each accept changes the mask the next candidate
is derived from -- so it is evaluated in speculative windows: the next W candidates are built
assuming the outcomes so far repeat (all-reject from the current mask, or an all-accept chain),
evaluated in one launch, and consumed up to and including the first misprediction.  Decisions,
returned masks and the REPORTED query count are exactly the reference's: only evaluations the
reference would have made are counted.  The images really classified (more, because of discarded
speculation) are counted by the evaluator (`TREvaluator.issued`; `attack_network` reports both).  An
evaluator with `speculate = False` (every duck-typed classifier) is never sent a speculative candidate.
"""
import importlib
import pickle
import random

import numpy as np
import torch

from . import parsearguments
from .coarse_reduction import get_coarse_reduced_mask
from .evaluator import TREvaluator, make_evaluator
from .transforms import draw_transforms

scoremod = None
FINE_WINDOW = 16          # largest speculative window (candidates per launch) in the fine reduction
FINE_WINDOW_MIN = 3       # the window doubles after a fully consumed launch and halves after a mis-speculation
# (16, 3): measured on config 2 (scripts/window_sweep.py, profiles/r03_window_sweep.txt: 0.360 s against 0.369 s at the
# earlier (32, 4)) and what a replay of the recorded accept / reject sequence under a launch-cost model predicts


def _storage_from_bytes(b):
    """Stand-in for torch.storage._load_from_bytes (what a pickled tensor's storage names): the real one is
    torch.load(..., weights_only=False), i.e. a second, unrestricted unpickler.  This one can only build storages."""
    import io
    return torch.load(io.BytesIO(b), weights_only=True)


class _HeatmapUnpickler(pickle.Unpickler):
    """Loader for the heatmap cache in the reference's format: pickle.dump of {'heatmap_query_ct', 'patches', 'indices',
    'order'} = an int, CPU torch tensors, tuples, one numpy index array (generate_mask.py:72-79).  The reference reloads it
    with plain pickle.load (:82-83); a pickle is a program, so here only the constructors those types need can be
    resolved, the tensor storages go through torch's weights-only loader, and any other global in the file raises
    instead of running."""
    _PLAIN = {("collections", "OrderedDict"), ("torch._utils", "_rebuild_tensor_v2"),
              ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
              ("numpy", "ndarray"), ("numpy", "dtype")}

    def find_class(self, module, name):
        if (module, name) == ("torch.storage", "_load_from_bytes"):
            return _storage_from_bytes
        if (module, name) in self._PLAIN:
            return super().find_class(module, name)
        raise pickle.UnpicklingError("heatmap file refers to %s.%s, which a heatmap cache never contains" % (module, name))


def load_heatmap(path):
    with open(path, 'rb') as f:
        heatmap = _HeatmapUnpickler(f).load()
    if not isinstance(heatmap, dict) or not {'heatmap_query_ct', 'patches', 'indices', 'order'} <= set(heatmap):
        raise ValueError("not a heatmap cache: " + str(path))
    return heatmap


def _load_scoremod(scorefile):
    name = scorefile.strip('.py')          # the reference's (character-set) strip, generate_mask.py:32
    importlib.invalidate_caches()
    try:
        return importlib.import_module(name)
    except ModuleNotFoundError:
        return importlib.import_module("graphite_amd." + name)


def generate_mask(model, img_v_small, img_v, lbl_v, mask, img_t_small, img_t, lbl_t, pt_file, scorefile, HEATMAP_MODE,
                  COARSE_ACCEPT_THRESHOLD, FINE_REJECT_THRESHOLD, num_xforms=100, net_size=32, model_type='GTSRB',
                  patch_size=4, stride_factor=4, heatmap_file=None, heatmap_out_path=None, max_mask_size=-1,
                  init_theta=None, _evaluator=None, _args=None):
    global scoremod
    scoremod = _load_scoremod(scorefile)
    args = _args or parsearguments.getarguments()

    print("Calculating Heatmap...")
    xforms = draw_transforms(num_xforms, model_type)      # 'PATCHGUARD': the baseline pg table (pg_generate_mask.py:43)
    ev = make_evaluator(_evaluator, model, img_v, xforms, pt_file, model_type, lbl_v, lbl_t)
    query_count = 0
    object_size = mask.sum() / 3
    patch = torch.ones((3, patch_size, patch_size))
    patches, indices = [], []

    if heatmap_file is None:
        for i in range(0, mask.size()[1] - patch_size, patch_size // stride_factor):
            for j in range(0, mask.size()[2] - patch_size, patch_size // stride_factor):
                new_mask = torch.zeros(mask.size())
                new_mask[:, i:min(i + patch_size, mask.size()[1]), j:min(j + patch_size, mask.size()[2])] = patch
                new_mask = new_mask * mask
                if torch.sum(new_mask) > 0:
                    patches.append(new_mask)
                    indices.append((i, j))
        print("num patches: ", len(patches))
        tr_scores, heatmap_query_ct = survey_heatmap(mask, patches, indices, img_v, img_t, img_v_small, img_t_small, lbl_t,
                                                     lbl_v, model, xforms, pt_file, net_size, HEATMAP_MODE, plot=True,
                                                     model_type=model_type, init_theta=init_theta, _evaluator=ev)
        query_count += heatmap_query_ct
        order = np.asarray(tr_scores).argsort()
        patches = [patches[ind] for ind in order]
        indices = [indices[ind] for ind in order]
        if heatmap_out_path is not None:
            with open(heatmap_out_path, 'wb') as f:
                pickle.dump({'heatmap_query_ct': heatmap_query_ct, 'patches': patches, 'indices': indices,
                             'order': order}, f)
    else:
        heatmap = load_heatmap(heatmap_file)
        heatmap_query_ct = heatmap['heatmap_query_ct']
        patches, indices, order = heatmap['patches'], heatmap['indices'], heatmap['order']
        query_count += heatmap_query_ct
    print("Heatmap completed.")

    coarse_red_mode = args.coarse_mode
    if coarse_red_mode != 'none':
        best_score, best_tr, best_mask, coarse_ct, pivot, patches, indices = get_coarse_reduced_mask(
            mask, object_size, patches, indices, img_v, img_t, lbl_v, lbl_t, model, xforms, pt_file, net_size,
            num_xforms=num_xforms, patch_size=patch_size, err_threshold=COARSE_ACCEPT_THRESHOLD,
            coarse_red_mode=coarse_red_mode, direction="forward", args=args, model_type=model_type,
            init_theta=init_theta, _evaluator=ev)
        query_count += coarse_ct
    else:
        best_mask = mask
    coarse_red_nbits = best_mask.sum() / 3

    print("Starting fine grained reduction")
    if max_mask_size > 0:
        # generate_mask.py:117-126: every pass works on COPIES of the patch list, lbd grows by 5, the passes' queries add up.
        # pg_generate_mask.py:119-131 (the older twin the PatchGuard driver kept): the passes share ONE list, which the
        # first pass consumes; lbd grows by 2.5; only the last pass's queries are reported.
        pg = model_type == 'PATCHGUARD'
        lbd = 5
        reduction_query_ct = 0
        while best_mask.sum() / 3 > max_mask_size:
            if pg and not patches and reduction_query_ct:
                break        # the reference would spin here for ever: its shared list is empty and the mask cannot shrink
            best_score, best_tr, best_mask, this_ct = get_fine_reduced_mask(
                best_mask, object_size, patches if pg else list(patches), indices if pg else list(indices), img_v,
                img_t_small, lbl_v, lbl_t, model, xforms,
                pt_file, net_size, err_threshold=FINE_REJECT_THRESHOLD, HEATMAP_MODE=HEATMAP_MODE, scoremod=scoremod,
                lbd=lbd, max_mask_size=max_mask_size, model_type=model_type, init_theta=init_theta, _evaluator=ev)
            lbd += 2.5 if pg else 5
            reduction_query_ct = this_ct if pg else reduction_query_ct + this_ct
    else:
        best_score, best_tr, best_mask, reduction_query_ct = get_fine_reduced_mask(
            best_mask, object_size, patches, indices, img_v, img_t_small, lbl_v, lbl_t, model, xforms, pt_file, net_size,
            err_threshold=FINE_REJECT_THRESHOLD, HEATMAP_MODE=HEATMAP_MODE, scoremod=scoremod, model_type=model_type,
            init_theta=init_theta, _evaluator=ev)
    reducer_nbits = best_mask.sum() / 3
    query_count += reduction_query_ct

    print('-' * 32)
    print("End of Mask Generation summary: queries", reduction_query_ct, "bits (coarse -> fine)", coarse_red_nbits.item(),
          reducer_nbits.item(), "tr", best_tr, "queries used", query_count)
    best_theta = (img_t_small - img_v_small) * best_mask
    attacked = ev.add_noise(img_v, best_mask, 1.0, best_theta)
    return attacked, query_count, best_mask, reducer_nbits.item(), best_tr


def get_heatmap(start_mask, patches, indices, img_v, img_t, lbl_v, lbl_t, model, xforms, pt_file, net_size, init_tr=0.5,
                plot=False, model_type='GTSRB', init_theta=None, _evaluator=None):
    ev = make_evaluator(_evaluator, model, img_v, xforms, pt_file, model_type, lbl_v, lbl_t)
    img_v_small = ev.resize(img_v, start_mask.size()[1], start_mask.size()[2])
    if img_t.size()[1] != start_mask.size()[1] or img_t.size()[2] != start_mask.size()[2]:
        img_t = ev.resize(img_t, start_mask.size()[1], start_mask.size()[2])
    if not patches:
        return [], 0
    # every leave-one-patch-out candidate in one launch; the same elementwise float32 products as the reference's loop
    # (generate_mask.py:168-172), taken over the stacked patches
    masks = start_mask[None] * (torch.ones(start_mask.size())[None] - torch.stack(patches))
    thetas = ((img_t - img_v_small) if init_theta is None else init_theta)[None] * masks
    res = ev.errors(masks, thetas)
    return [1 - e for e, _ in res], sum(n for _, n in res)


def get_fine_reduced_mask(start_mask, object_size, patches, indices, img_v, img_t, lbl_v, lbl_t, model, xforms, pt_file,
                          net_size, err_threshold=0.75, HEATMAP_MODE=None, scoremod=None, lbd=5, max_mask_size=-1,
                          model_type='GTSRB', init_theta=None, _evaluator=None):
    ev = make_evaluator(_evaluator, model, img_v, xforms, pt_file, model_type, lbl_v, lbl_t)
    reduction_query_ct = 0
    img_v_small = ev.resize(img_v, start_mask.size()[1], start_mask.size()[2])
    if img_t.size()[1] != img_v_small.size()[1] or img_t.size()[2] != img_v_small.size()[2]:
        img_t = ev.resize(img_t, start_mask.size()[1], start_mask.size()[2])

    def theta_of(m):
        return (img_t - img_v_small) * m if init_theta is None else init_theta * m

    theta = theta_of(start_mask)
    success_rate, query_ct = ev.error(start_mask, theta)
    reduction_query_ct += query_ct
    init_tr = 1 - success_rate
    print("init_tr", init_tr)
    print("init bits", start_mask.sum() / 3)
    best_score = scoremod.score_fn(theta + img_v_small, start_mask, success_rate, object_size, None,
                                   threshold=err_threshold, lbd=lbd)
    best_tr, best_mask = init_tr, start_mask
    ones = torch.ones(start_mask.size())
    last_accept = True                                   # first window speculates an accept chain
    # an evaluator that must not see candidates the sequential algorithm would not evaluate (a duck-typed classifier
    # behind .predict, whose query count is the attack's cost) gets windows of ONE: no speculation
    wmax = FINE_WINDOW if getattr(ev, "speculate", True) else 1
    wmin = min(wmax, FINE_WINDOW_MIN)
    win = max(1, min(wmax, 2 * wmin))

    done = False
    while patches and not done:
        # ---- build a speculative window from the END of the list (the reference pops from the end) ----
        spec_mask = best_mask
        window = []                                      # (position in list, patch, index, candidate mask, assumed base)
        pos = len(patches)
        while pos > 0 and len(window) < win:
            pos -= 1
            p = patches[pos]
            if torch.max(p * spec_mask) == 0:            # the reference would skip it (given the assumed outcomes)
                window.append((pos, p, indices[pos], None, spec_mask))
                continue
            cand = spec_mask * (ones - p)
            window.append((pos, p, indices[pos], cand, spec_mask))
            if last_accept:
                spec_mask = cand                         # assume accepted: the next candidate builds on it
        live = [w for w in window if w[3] is not None]
        live_thetas = {id(w[3]): theta_of(w[3]) for w in live}           # reused by the scoring below
        res = ev.errors([w[3] for w in live], [live_thetas[id(w[3])] for w in live]) if live else []
        res_it = iter(res)
        # ---- consume in the reference's order; stop at the first outcome that breaks the assumption ----
        consumed = 0
        for (wpos, p, idx, cand, base) in window:
            if base is not best_mask and not torch.equal(base, best_mask):
                break                                    # speculation no longer valid from here on
            consumed += 1
            if cand is None:
                continue
            success_rate, query_ct = next(res_it)
            reduction_query_ct += query_ct               # only evaluations the reference would have made
            theta = live_thetas[id(cand)]
            score = scoremod.score_fn(theta + img_v_small, cand, success_rate, object_size, idx, best_score=best_score,
                                      threshold=err_threshold, lbd=lbd)
            if score < best_score:
                best_score, best_mask, best_tr = score, cand, 1 - success_rate
                accepted = True
                if max_mask_size > 0 and best_mask.sum() / 3 < max_mask_size:
                    done = True
            else:
                accepted = False
            mispredicted = accepted != last_accept
            last_accept = accepted
            if done or mispredicted:
                break
        win = min(wmax, 2 * win) if consumed == len(window) else max(wmin, win // 2)
        del patches[len(patches) - consumed:]
        del indices[len(indices) - consumed:]
    return best_score, best_tr, best_mask, reduction_query_ct


def survey_heatmap(mask, patches, indices, img_v, img_t, img_v_small, img_t_small, lbl_t, lbl_v, model, xforms, pt_file,
                   net_size, HEATMAP_MODE, plot=False, model_type='GTSRB', init_theta=None, _evaluator=None):
    if HEATMAP_MODE == 'Random':
        return [random.random() for _ in range(len(patches))], 0
    if HEATMAP_MODE == 'Victim':
        ev = TREvaluator(model, img_t, xforms, pt_file, model_type, lbl_t, lbl_v) if _evaluator is None else _evaluator.swapped(img_t, lbl_t, lbl_v)
        return get_heatmap(mask, patches, indices, img_t, img_v_small, lbl_t, lbl_v, model, xforms, pt_file, net_size,
                           plot=plot, model_type=model_type, _evaluator=ev)
    return get_heatmap(mask, patches, indices, img_v, img_t_small, lbl_v, lbl_t, model, xforms, pt_file, net_size,
                       plot=plot, model_type=model_type, init_theta=init_theta, _evaluator=_evaluator)
