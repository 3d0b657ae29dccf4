"""Fused, batched transform-robustness evaluation (the entry the batched search drivers use).

`tr_hits(...)` == for every candidate c:
    imgs = get_transformed_images(image, masks[c], transforms, lbd, thetas[c], pt_file, 32, model_type)
    err, n = run_predictions(model, imgs, label, target)         # reference utils.py:60
    hits[c] = round((1 - err) * n)
but in one pass over C x T images on the GPU.  Sharding over ranks is OPT-IN (`distributed=True` or a process
group): the flattened (candidate x transform) index space is then split evenly over the ranks and the per-candidate
counts are summed with ONE int32[C] all-reduce (RCCL over xGMI on a GPU node, gloo in the CPU tests).  It is opt-in
because every rank of the group must make the same call with the same (C, T): a process that merely has a
torch.distributed group initialised for something else must not find collectives inside boost / generate_mask.
"""
import numpy as np
import torch

from . import _hip
from .transforms import build_xform_table, normalizes, XFORM_DTYPE


def shard_range(total, rank, world):
    """Contiguous, balanced slice [begin, begin+count) of `total` items for `rank` of `world`."""
    base, rem = divmod(int(total), int(world))
    begin = rank * base + min(rank, rem)
    return begin, base + (1 if rank < rem else 0)


def allreduce_hits(hits, group=None):
    """Sum the per-candidate partial counts over the ranks (no-op without torch.distributed).
    With the nccl (= RCCL) backend the int32[C] vector is reduced in place on the GPU over xGMI;
    with gloo (CPU rehearsal / tests) it takes a round trip through host memory."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if dist.get_backend(group) == "gloo" and hits.is_cuda:
            h = hits.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
            hits.copy_(h)
        else:
            dist.all_reduce(hits, op=dist.ReduceOp.SUM, group=group)
    return hits


def _assert_same_work(total, hits, group):
    """Before EVERY sharded reduction: one MAX all-reduce of (total, C, -total, -C) proves that every rank is evaluating
    the same index space (ranks that disagree would otherwise sum unrelated counts, or pair a C-long reduction with a
    C'-long one).  It is a fixed-size int64[4] collective, so ranks that disagree still meet in it and all of them raise;
    it is not cached per shape -- a cache would let rank 0 skip the check that rank 1 is waiting in."""
    import torch.distributed as dist
    v = torch.tensor([total, hits.numel(), -total, -hits.numel()], dtype=torch.int64)
    if dist.get_backend(group) != "gloo":
        v = v.to(hits.device)
    dist.all_reduce(v, op=dist.ReduceOp.MAX, group=group)
    v = v.cpu().tolist()
    if not (v[0] == -v[2] == total and v[1] == -v[3] == hits.numel()):
        raise RuntimeError("sharded tr_hits: ranks disagree on the work (images max %d / min %d, candidates max %d / min %d)"
                           % (v[0], -v[2], v[1], -v[3]))


def sharded_hits(eval_slice, total, distributed=False, group=None):
    """Run `eval_slice(begin, count) -> int32[C]` (partial counts of images [begin, begin+count) of the
    flattened candidate x transform index) on this rank's slice and sum over the ranks.  Unsharded unless
    `distributed` is true (or a group is given) AND torch.distributed is initialised with more than one rank."""
    import torch.distributed as dist
    sharded = ((distributed or group is not None) and dist.is_available() and dist.is_initialized()
               and dist.get_world_size(group) > 1)
    if not sharded:
        return eval_slice(0, total)
    begin, count = shard_range(total, dist.get_rank(group), dist.get_world_size(group))
    hits = eval_slice(begin, count)
    _assert_same_work(total, hits, group)
    return allreduce_hits(hits, group)


def _as_table(transforms, w, h, pt_file):
    if isinstance(transforms, _hip.XformTable):
        return transforms
    if isinstance(transforms, np.ndarray) and transforms.dtype == XFORM_DTYPE:
        return transforms
    return build_xform_table(transforms, w, h, pt_file)


def tr_hits(model, image, masks, thetas, transforms, target, pt_file='inputs/GTSRB/Points/14.csv', lbd=1.0,
            model_type='GTSRB', distributed=False, return_preds=False, group=None):
    """hits[c] = number of the T transformed images of candidate c classified as `target`.

    image [3,H,W]; masks, thetas [C,3,h,w] (CPU or CUDA tensors / numpy); transforms = the list from
    get_transform_params, a table from build_xform_table, or a cached _hip.XformTable.
    Returns an int32 CUDA tensor [C] (and int64 predictions [C,T] if return_preds; with sharding
    each rank's predictions cover its own slice, -1 elsewhere).
    distributed / group: shard over the ranks of torch.distributed's default (or the given) group -- every rank must
    make this same call; off by default.
    """
    hip_model = getattr(model, "hip_model", model)
    img = _hip.to_dev(image)
    tab = _as_table(transforms, img.shape[2], img.shape[1], pt_file)
    m, t = _hip.to_dev(masks), _hip.to_dev(thetas)
    if m.dim() == 3:
        m, t = m[None], t[None]
    xf = _hip.xform_table(tab)
    total = m.shape[0] * xf.T
    keep = {}

    def eval_slice(begin, count):
        hits, preds, _ = _hip.tr_hits_dev(hip_model, img, m, t, xf, float(lbd), normalizes(model_type), int(target),
                                          begin, count, want_preds=return_preds)
        keep["preds"] = preds
        return hits

    hits = sharded_hits(eval_slice, total, distributed, group)
    if return_preds:
        return hits, keep["preds"].view(m.shape[0], xf.T)
    return hits


def tr_error_rates(model, image, masks, thetas, transforms, target, **kw):
    """[(error_rate, queries)] per candidate, in run_predictions' convention (utils.py:57)."""
    hits = tr_hits(model, image, masks, thetas, transforms, target, **kw).cpu().numpy()
    T = len(transforms) if not isinstance(transforms, _hip.XformTable) else transforms.T
    return [(1.0 - int(h) * 1.0 / T, T) for h in hits]


def eot_gradient(model, image, adv, mask, transforms, target, pt_file='inputs/GTSRB/Points/14.csv', normalize=True):
    """One step of the whitebox EOT loop (reference whitebox/whitebox_attack.py:149-162), everything on the GPU:
    x_t = transform_wb(image, adv, mask, transform_t) [- 0.5]; loss_t = CrossEntropy(model(x_t), target);
    returns (mean over t of d loss_t / d adv  [3,H,W] CUDA, logits [T,ncls] CUDA).
    Three launches groups: gr_wb_forward, gr_model_input_grad, gr_wb_backward."""
    import torch
    from . import transforms as T
    hip_model = getattr(model, "hip_model", model)
    o, a, m = _hip.to_dev(T._chw3(image)), _hip.to_dev(T._chw3(adv)), _hip.to_dev(T._chw3(mask))
    if isinstance(transforms, np.ndarray) and transforms.dtype == _hip.WB_XFORM_DTYPE:
        tab = transforms
    else:
        tab = T.build_wb_table(transforms, o.shape[2], o.shape[1], pt_file)
    x = _hip.wb_forward(o, a, m, tab)
    if normalize:
        x = x - 0.5
    logits, gx = hip_model.input_grad(x, target)
    g = _hip.wb_backward(o, a, m, tab, gx)
    return g / float(len(tab)), logits
