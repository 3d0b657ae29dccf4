"""run_predictions -- drop-in for the reference's `utils.py`.

  run_predictions_untargeted  utils.py:6-26
  run_predictions_targeted    utils.py:29-57
  run_predictions             utils.py:60-68

Return values are the reference's: `(1 - hits/len(imgs), len(imgs))`, i.e. the ERROR rate (not the
transform robustness) and the number of classifier queries.  The model is duck-typed, as in the
reference: anything with `.predict(batch) -> LongTensor[B]`.  Batches are 256 images, the
reference's batch size, so within ONE run_predictions call a stateful / order-sensitive model sees the
reference's predict() calls; a graphite_amd model classifies each batch on the GPU (predict() uploads once,
returns device labels).  Across calls the search drivers keep the reference's sequence of evaluations for
such a model too: they speculate (evaluate candidates the sequential algorithm may discard) only through
the fused HIP evaluator (`TREvaluator.speculate`), never through a duck-typed `.predict`.
"""
import torch

BATCH = 256


def _batches(imgs):
    """Yield float32 tensors [<=256, C, H, W]; an image larger than the first is cropped to it and a
    smaller one zero-padded (utils.py:42-44)."""
    c, h, w = imgs[0].size()[0], imgs[0].size()[1], imgs[0].size()[2]
    for s in range(0, len(imgs), BATCH):
        part = imgs[s:s + BATCH]
        if all(tuple(i.size()) == (c, h, w) for i in part):
            yield torch.stack([i.detach().cpu().float() for i in part])
        else:
            t = torch.zeros((len(part), c, h, w))
            for k, i in enumerate(part):
                my, mx = min(i.size()[1], h), min(i.size()[2], w)
                t[k, :, :my, :mx] = i[:, :my, :mx]
            yield t


def _count(preds, label, equal):
    preds = torch.as_tensor(preds)
    lbl = torch.full_like(preds, int(label))
    return int(((preds == lbl) if equal else (preds != lbl)).sum().item())


def run_predictions_untargeted(model, imgs, label):
    n = 0
    for b in _batches(imgs):
        n += _count(model.predict(b), label, equal=False)
    return (1.0 - n * 1.0 / len(imgs)), len(imgs)


def run_predictions_targeted(model, imgs, target, victim=None):
    n = 0
    for b in _batches(imgs):
        n += _count(model.predict(b), target, equal=True)
    return (1.0 - n * 1.0 / len(imgs)), len(imgs)


def run_predictions(model, imgs, label, target=None):
    """imgs: transformed images; label: the victim's label; target: the target label (None = untargeted)."""
    if target is None:
        return run_predictions_untargeted(model, imgs, label)
    return run_predictions_targeted(model, imgs, target, label)
