"""TR evaluation service used by the batched search drivers (boost / generate_mask / coarse_reduction).

One `TREvaluator` = one (model, victim image, transform list, labels) context, i.e. exactly what is
constant across the ~1,700 `get_transformed_images` + `run_predictions` pairs of an attack.
`errors(masks, thetas)` evaluates any number of candidates in ONE pass and returns, per candidate,
the reference's `(1 - hits/N, N)` pair (utils.py:26,57), so callers keep the reference's arithmetic
and query accounting.  Image-space helpers (`resize`, `add_noise`) are here too so that the drivers
have a single seam: the CPU tests substitute an evaluator backed by the oracle / a table.
"""
import numpy as np
import torch

from . import _hip
from .transforms import build_xform_table


class TREvaluator:
    def __init__(self, model, image, transforms, pt_file, model_type='GTSRB', label=None, target=None, lbd=1.0,
                 distributed=False, group=None):
        """distributed / group: shard every evaluation over torch.distributed's ranks (engine.tr_hits); opt-in, and only
        valid when ALL ranks run the identical seeded driver in lock-step."""
        self.model, self.model_type, self.label, self.target, self.lbd = model, model_type, label, target, float(lbd)
        self.distributed, self.group = bool(distributed), group
        self.image = image
        self.n = len(transforms)
        self.transforms = transforms
        self.pt_file = pt_file
        self._img_d = _hip.to_dev(image)
        self._xf = _hip.xform_table(build_xform_table(transforms, image.shape[2], image.shape[1], pt_file)) if self.n else None
        self._fused = hasattr(model, "hip_model")
        # Speculative batching (fine-reduction windows, boost's step + next directions in one launch) evaluates candidates
        # the sequential algorithm may never ask for.  That is free for the fused HIP classifier; a duck-typed model
        # (anything with .predict) must see exactly the reference's predict() calls in the reference's order, so the
        # drivers speculate only when this is true.
        self.speculate = self._fused
        self.issued = 0          # images really sent to the classifier (>= the reference-equivalent query count)

    # ---- candidates -> [(error_rate, queries)] -------------------------------------------------
    def errors(self, masks, thetas):
        masks = torch.stack([m.float() for m in masks]) if isinstance(masks, (list, tuple)) else masks
        thetas = torch.stack([t.float() for t in thetas]) if isinstance(thetas, (list, tuple)) else thetas
        C = masks.shape[0]
        if C == 0:
            return []
        N = self.n
        self.issued += C * max(N, 1)
        if self._fused and N > 0:
            from .engine import tr_hits
            if self.target is not None:
                hits = tr_hits(self.model, self._img_d, masks, thetas, self._xf, self.target, lbd=self.lbd,
                               model_type=self.model_type, distributed=self.distributed, group=self.group).cpu().numpy()
                return [((1.0 - int(h) * 1.0 / N), N) for h in hits]
            same = tr_hits(self.model, self._img_d, masks, thetas, self._xf, self.label, lbd=self.lbd,
                           model_type=self.model_type, distributed=self.distributed, group=self.group).cpu().numpy()
            return [((1.0 - (N - int(h)) * 1.0 / N), N) for h in same]          # untargeted: count preds != label
        # duck-typed classifier (anything with .predict): images from the fused warp, labels from the model
        from .transforms import get_transformed_images
        from .utils import run_predictions
        out = []
        for c in range(C):
            imgs = get_transformed_images(self.image, masks[c], self.transforms, self.lbd, thetas[c], self.pt_file,
                                          model_type=self.model_type)
            out.append(run_predictions(self.model, imgs, self.label, self.target))
        return out

    def error(self, mask, theta):
        return self.errors(mask[None], theta[None])[0]

    def swapped(self, image, label, target):
        """Same model / transforms with another base image and labels (the 'Victim' heatmap,
        generate_mask.py:262-263, transforms the TARGET image and counts VICTIM-label hits)."""
        return TREvaluator(self.model, image, self.transforms, self.pt_file, self.model_type, label, target, self.lbd,
                           self.distributed, self.group)

    # ---- image helpers (cv2.resize / add_noise in the reference's drivers) ----------------------
    @staticmethod
    def resize(t, h, w):
        """cv2.resize of a CHW tensor to (h, w): boost.py:33,42; generate_mask.py:153; coarse_reduction.py:20."""
        return torch.from_numpy(_hip.resize_linear(np.ascontiguousarray(t.detach().cpu().numpy(), dtype=np.float32), h, w))

    @staticmethod
    def add_noise(image, mask, lbd, theta, clip=True):
        from .transforms import add_noise
        return add_noise(image, mask, lbd, theta, clip=clip)


def make_evaluator(spec, model, image, transforms, pt_file, model_type, label, target):
    """`spec` is None (GPU evaluator), an evaluator object, or a factory with TREvaluator's signature
    (tests use a factory to bind an oracle-backed evaluator to the transform list drawn inside a driver)."""
    if spec is None:
        return TREvaluator(model, image, transforms, pt_file, model_type, label, target)
    if hasattr(spec, "errors") or hasattr(spec, "error"):
        return spec
    return spec(model, image, transforms, pt_file, model_type, label, target)
