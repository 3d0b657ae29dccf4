"""Evaluator seam (graphite_amd.evaluator) backed entirely by the CPU oracle -- TEST INFRASTRUCTURE.
Used to check that a whole attack run on the GPU follows the trajectory the oracle dictates."""
import numpy as np
import torch

import oracle_lib as O
from graphite_amd import weights
from graphite_amd.transforms import build_xform_table, normalizes


class OracleEvaluator:
    def __init__(self, state_dict, image, transforms, pt_file, model_type, label, target, nthreads=16):
        self.state_dict, self.transforms, self.pt_file, self.model_type = state_dict, transforms, pt_file, model_type
        self.tensors = self.tensor_list(state_dict)
        self.image = np.ascontiguousarray(image.numpy(), dtype=np.float32)
        self.tab = build_xform_table(transforms, image.shape[2], image.shape[1], pt_file)
        self.T, self.label, self.target, self.normalize = len(transforms), label, target, normalizes(model_type)
        self.nthreads = nthreads
        self.evaluated = 0

    @staticmethod
    def tensor_list(state_dict):
        return weights.gtsrb_tensor_list(state_dict)

    @classmethod
    def factory(cls, state_dict, nthreads=16):
        return lambda model, image, transforms, pt_file, model_type, label, target: cls(
            state_dict, image, transforms, pt_file, model_type, label, target, nthreads)

    def errors(self, masks, thetas):
        masks = np.stack([np.asarray(m, np.float32) for m in masks])
        thetas = np.stack([np.asarray(t, np.float32) for t in thetas])
        self.evaluated += len(masks)
        px = O.warp_batch(self.image, masks, thetas, self.tab, normalize=self.normalize, nthreads=self.nthreads)
        hits = (self.labels(px.reshape(-1, 3, 32, 32)).reshape(len(masks), self.T) == self.target).sum(1)
        return [((1.0 - int(h) * 1.0 / self.T), self.T) for h in hits]

    def labels(self, x):
        lg = O.gtsrb_forward(self.tensors, x, nthreads=self.nthreads)
        return O.argmax_hits(lg, self.T, self.target)[0]

    def error(self, mask, theta):
        return self.errors([mask], [theta])[0]

    def swapped(self, img, label, target):
        """The 'Victim' heatmap (generate_mask.py:262-263 in the reference): same model / transforms on the TARGET image."""
        return type(self)(self.state_dict, img, self.transforms, self.pt_file, self.model_type, label, target,
                          self.nthreads)

    @staticmethod
    def resize(t, h, w):
        return torch.from_numpy(O.resize_linear(np.ascontiguousarray(t.numpy(), dtype=np.float32), h, w))

    @staticmethod
    def add_noise(image, mask, lbd, theta, clip=True):
        return torch.from_numpy(O.add_noise(image.numpy(), mask.numpy(), theta.numpy(), lbd, clip)[0])


def patchguard_oracle_evaluator(arch, size, window=6):
    """OracleEvaluator whose classifier is the PatchGuard restatement (bicubic -> BagNet -> masking_defense)."""
    class OraclePGEvaluator(OracleEvaluator):
        @staticmethod
        def tensor_list(state_dict):
            return weights.bagnet_tensor_list(state_dict, arch["layers"])

        def labels(self, x):
            lo = O.bagnet_forward(self.tensors, O.pg_preprocess(x, size, nthreads=self.nthreads), nthreads=self.nthreads, **arch)
            return O.masking_defense(lo, min(window, lo.shape[1]), 0.0)[1]
    return OraclePGEvaluator
